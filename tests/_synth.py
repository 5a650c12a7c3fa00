"""Seeded synthetic inputs of SURVEY.md section 8(d) for the parity tests (CPU tensors)."""
import torch


def make_head_inputs(N, H, W, C, K, M=1, Ca=None, seed=42, w_std=None, bias_std=0.1,
                     dtype=torch.float32, relu_x=True):
    g = torch.Generator().manual_seed(seed)
    Ca = C if Ca is None else Ca
    w_std = (1.0 / C ** 0.5) if w_std is None else w_std
    X = torch.randn(N, H, W, C, generator=g)
    if relu_x:
        X = torch.relu(X)           # conv5 is post-ReLU (resnet_v1.py:108)
    Xatt = X if Ca == C else torch.relu(torch.randn(N, H, W, Ca, generator=g))
    Wa = torch.randn(Ca, M, generator=g) * (1.0 / Ca ** 0.5 if w_std != 1e-3 else 1e-3)
    ba = torch.randn(M, generator=g) * bias_std
    Wt = torch.randn(C, K, generator=g) * w_std
    bt = torch.randn(K, generator=g) * bias_std
    labels = torch.randint(0, K, (N,), generator=g)
    X = X.to(dtype)
    Xatt = X if Ca == C else Xatt.to(dtype)       # fused path: Xatt IS X (same object)
    return dict(X=X, Xatt=Xatt, Wa=Wa, ba=ba, Wt=Wt, bt=bt, labels=labels)
