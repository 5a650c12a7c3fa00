"""CPU oracle -- test infrastructure only.

Parity status (details in each module's header): attn_pool_oracle.py is PINNED to the reference's own head /
loss graph code (tests/golden/make_head_reference.py executes nets_factory.py, loss.py, config.py and the
arg-scopes behind a TF stand-in; 1e-12 in tests/test_reference_fixtures_cpu.py); labels_eval_oracle.py is pinned
for everything after the raster canvas (the reference's train_preprocess_pipeline, 14 cases) and for the mAP
functions (the reference's cap_eval_utils.py), while cv::circle / cv::GaussianBlur themselves are restated from
OpenCV's published algorithm -- OpenCV is not in this image.  Not pinnable here: TensorFlow's fp32 kernel
rounding and RNG streams.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
