"""autograd._flush_heads (round 5): how the recorded backward calls of the xyz heads are grouped into multi-head launches
(ops.mlp128_bwd_heads) — by input tensor and scale, never two heads that add into the same gradient buffer, at most
MLP128_MAX_HEADS per launch — checked without a GPU by standing in for the launch."""
import torch


def _head(xyz, scale, buf=None):
    dks = [buf if buf is not None else torch.zeros(3)] + [torch.zeros(2) for _ in range(4)]
    return dict(xyz=xyz, dout=torch.zeros(4, 3), blob=torch.zeros(8, dtype=torch.uint8), dks=dks, dbs=[torch.zeros(2) for _ in range(5)],
                out_act=None, xyz_scale=scale, post_scale=1.0)


def test_recorded_heads_are_grouped_into_launches(nfx_lib, monkeypatch):
    from nerfactor_amd import autograd, ops
    launches = []
    monkeypatch.setattr(ops, 'mlp128_bwd_heads', lambda kind, xyz, heads, xyz_scale=1., prec='bf16': launches.append((xyz, xyz_scale, heads)))
    a, b = torch.zeros(4, 3), torch.zeros(4, 3)
    shared = torch.zeros(3)
    pending = [_head(a, 1.0), _head(a, 1.0), _head(b, 1.0), _head(a, 0.5), _head(a, 1.0, shared), _head(a, 1.0, shared),
               _head(a, 1.0), _head(a, 1.0)]
    autograd._heads['pending'], autograd._heads['armed'] = list(pending), True
    autograd._flush_heads()
    assert autograd._heads['pending'] == [] and autograd._heads['armed'] is False
    sizes = [(x is a, s, len(h)) for x, s, h in launches]
    # first launch: the heads over `a` at scale 1 with distinct buffers, at most MLP128_MAX_HEADS (= 4) of them ...
    assert sizes[0] == (True, 1.0, ops.MLP128_MAX_HEADS)
    assert sum(n for _, _, n in sizes) == len(pending)
    for x, s, heads in launches:
        ptrs = [h[2][0].data_ptr() for h in heads]
        assert len(set(ptrs)) == len(ptrs)                      # ... never two heads adding into one buffer in a launch
    assert any(x is b for x, _, _ in launches) and any(s == 0.5 for _, s, _ in launches)
    # every recorded head was launched exactly once, in some launch with ITS input and scale
    seen = [(id(x), s, h[2][0].data_ptr()) for x, s, heads in launches for h in heads]
    want = [(id(p['xyz']), p['xyz_scale'], p['dks'][0].data_ptr()) for p in pending]
    assert sorted(seen) == sorted(want)


def test_a_forward_drops_heads_left_by_a_failed_backward(nfx_lib, monkeypatch):
    from nerfactor_amd import autograd, ops
    monkeypatch.setattr(ops, 'mlp128_xyz_fwd', lambda xyz, blob, out_dim, **kw: torch.zeros(xyz.shape[0], out_dim))
    autograd._heads['pending'], autograd._heads['armed'] = [_head(torch.zeros(4, 3), 1.0)], True
    autograd.Mlp128Xyz.apply(torch.zeros(4, 3), None, lambda: None, 'bf16', 3, None, 1.0, 1.0, 0.0)
    assert autograd._heads['pending'] == [] and autograd._heads['armed'] is False


def test_a_forward_inside_a_running_backward_keeps_the_recorded_heads(nfx_lib, monkeypatch):
    """a forward of the op that runs WHILE a backward pass is in flight (activation re-computation, a second
    model) must not drop the heads that pass has already recorded — their parameter gradients would be lost silently."""
    from nerfactor_amd import autograd, ops
    monkeypatch.setattr(ops, 'mlp128_xyz_fwd', lambda xyz, blob, out_dim, **kw: torch.zeros(xyz.shape[0], out_dim))
    recorded = [_head(torch.zeros(4, 3), 1.0)]
    seen = {}

    class Recompute(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            autograd._heads['pending'], autograd._heads['armed'] = list(recorded), True
            autograd.Mlp128Xyz.apply(torch.zeros(4, 3), None, lambda: None, 'bf16', 3, None, 1.0, 1.0, 0.0)   # a forward mid-backward
            seen['pending'] = list(autograd._heads['pending'])
            autograd._heads['pending'], autograd._heads['armed'] = [], False
            return g

    x = torch.ones(3, requires_grad=True)
    Recompute.apply(x).sum().backward()
    assert seen['pending'] == recorded
