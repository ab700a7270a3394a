"""Helpers shared by the GPU parity tests."""
import torch

DEV = "cuda"
BF = torch.bfloat16


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def assert_close(name, got, ref, tol):
    e = rel_l2(got, ref)
    assert e <= tol, f"{name}: rel-L2 {e:.3e} > {tol:.1e} (|ref|={ref.float().norm().item():.3e})"
    assert torch.isfinite(got.float()).all(), f"{name}: non-finite values"
    return e


def bf(x):
    """Round an fp32 tensor to bf16 and move it to the GPU (test inputs are bf16-exact so that the fp32
    reference sees the same numbers as the kernel)."""
    return x.to(BF).to(DEV)


def load_from_oracle(product, oracle):
    """Copy the oracle's weights into the product module (rank_output aliases itm_score.fc rows, skip it)."""
    sd = {k: v for k, v in oracle.state_dict().items() if not k.startswith("rank_output.")}
    missing, unexpected = product.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("rank_output.") for k in missing), missing
    return product
