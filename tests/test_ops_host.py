"""Host-side logic of fiber_amd/ops.py that needs no GPU: the one-shot hand-over of column sums from a producing backward
(window attention: the qkv bias gradient) to the linear backward that follows it."""
import torch

from fiber_amd import ops


class _Producer(torch.autograd.Function):
    """Stands in for the window-attention backward: returns a gradient and offers `sums` for it."""

    @staticmethod
    def forward(ctx, x, sums, box):
        ctx.sums, ctx.box = sums, box
        return x.clone()

    @staticmethod
    def backward(ctx, dy):
        g = dy.clone()
        ops._offer_colsum(g.view(-1, g.shape[-1]), ctx.sums)
        ctx.box.append(g)
        return g, None, None


class _Consumer(torch.autograd.Function):
    """Stands in for a linear backward: takes the slot for the dY it receives."""

    @staticmethod
    def forward(ctx, x, box):
        ctx.box = box
        return x * 2.0

    @staticmethod
    def backward(ctx, dy):
        ctx.box.append(ops._take_colsum(dy.view(-1, dy.shape[-1])))
        return dy * 2.0, None


def test_colsum_handover_matches_only_the_offered_tensor():
    sums = torch.arange(6.0)
    # producer's gradient flows straight into the consumer: the consumer gets the offered sums
    x = torch.randn(4, 6, requires_grad=True)
    got, grads = [], []
    _Producer.apply(_Consumer.apply(x, got), sums, grads).sum().backward()
    assert len(got) == 1 and got[0] is sums
    assert getattr(ops._hint_tls, "slot", None) is None
    # nobody takes the offer: it does not outlive the autograd pass
    x = torch.randn(4, 6, requires_grad=True)
    _Producer.apply(x, sums, []).sum().backward()
    assert getattr(ops._hint_tls, "slot", None) is None
    # a consumer whose dY is another tensor (different storage) gets nothing and clears the slot
    x = torch.randn(4, 6, requires_grad=True)
    got = []
    y = _Consumer.apply(x, got)
    z = _Producer.apply(y * 1.0, sums, [])          # the multiplication puts a fresh tensor between producer and consumer
    z.sum().backward()
    assert got == [None] and getattr(ops._hint_tls, "slot", None) is None
    # same storage but another shape is not a match either
    t = torch.zeros(4, 6)

    class _Offer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a):
            return a.clone()

        @staticmethod
        def backward(ctx, dy):
            ops._offer_colsum(t, sums)
            res.append(ops._take_colsum(t.view(2, 12)))
            res.append(ops._take_colsum(t))            # the first take cleared the slot
            return dy

    res = []
    _Offer.apply(torch.randn(3, requires_grad=True)).sum().backward()
    assert res == [None, None]
