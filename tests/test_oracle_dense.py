"""Pins the CPU oracle's sparse operators against torch-CPU dense ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import scn_ref as scn


def _random_sparse(seed, S=12, B=2, n=150, C=5, dup=True):
    g = np.random.default_rng(seed)
    coords = np.concatenate([g.integers(0, S, (n, 3)), g.integers(0, B, (n, 1))], 1).astype(np.int64)
    if dup:
        coords = np.concatenate([coords, coords[g.integers(0, n, n // 4)]], 0)
    feats = torch.from_numpy(g.standard_normal((len(coords), C)).astype(np.float32))
    return torch.from_numpy(coords), feats


def _densify(feat, level, S, B):
    d = torch.zeros(B, feat.shape[1], S, S, S)
    c = level.coords
    d[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]] = feat
    return d


def _sample(dense, level):
    c = level.coords
    return dense[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]]


def test_input_layer_first_occurrence_and_mean():
    coords = torch.tensor([[1, 2, 3, 0], [4, 4, 4, 0], [1, 2, 3, 0], [1, 2, 3, 1], [4, 4, 4, 0]])
    feats = torch.tensor([[1.0], [10.0], [3.0], [7.0], [20.0]])
    f, level, p2v = scn.input_layer(coords, feats, 16, 4)
    assert p2v.tolist() == [0, 1, 0, 2, 1]
    assert level.coords.tolist() == [[1, 2, 3, 0], [4, 4, 4, 0], [1, 2, 3, 1]]
    assert torch.allclose(f, torch.tensor([[2.0], [15.0], [7.0]]))


@pytest.mark.parametrize("seed", [0, 1])
def test_subm_equals_dense_conv3d(seed):
    S, B, Cin, Cout = 12, 2, 5, 7
    coords, feats = _random_sparse(seed, S, B, C=Cin)
    f, level, _ = scn.input_layer(coords, feats, S, 4)
    w = torch.randn(27, Cin, Cout)
    out = scn.rule_conv(f, w, scn.subm_rulebook(level), level.n)
    wt = w.view(3, 3, 3, Cin, Cout).permute(4, 3, 0, 1, 2).contiguous()
    ref = _sample(F.conv3d(_densify(f, level, S, B), wt, padding=1), level)
    assert torch.allclose(out, ref, atol=1e-4, rtol=1e-4)
    rb = level.subm
    assert rb.n_rules == rb.offsets[-1] and np.array_equal(rb.bucket(13)[0], np.arange(level.n))
    for k in range(27):  # canonical order (iii): sorted by out inside a bucket
        assert np.all(np.diff(rb.bucket(k)[1]) > 0)


@pytest.mark.parametrize("seed", [0, 3])
def test_conv_deconv_equal_dense(seed):
    S, B, Cin, Cout = 12, 2, 4, 6
    coords, feats = _random_sparse(seed, S, B, C=Cin)
    f, level, _ = scn.input_layer(coords, feats, S, 4)
    rb, coarse = scn.down_rulebook(level)
    # canonical order (ii): coarse ids = first occurrence of parents scanning fine ids
    parents = [tuple(r) for r in np.concatenate([level.coords[:, :3] >> 1, level.coords[:, 3:]], 1)]
    seen = list(dict.fromkeys(parents))
    assert [tuple(r) for r in coarse.coords] == seen
    w = torch.randn(8, Cin, Cout)
    out = scn.rule_conv(f, w, rb, coarse.n)
    wt = w.view(2, 2, 2, Cin, Cout).permute(4, 3, 0, 1, 2).contiguous()
    ref = _sample(F.conv3d(_densify(f, level, S, B), wt, stride=2), coarse)
    assert torch.allclose(out, ref, atol=1e-4, rtol=1e-4)
    # deconvolution back onto the fine active set
    w2 = torch.randn(8, Cout, Cin)
    up = scn.rule_conv(out, w2, rb, level.n, transpose_roles=True)
    wt2 = w2.view(2, 2, 2, Cout, Cin).permute(3, 4, 0, 1, 2).contiguous()
    ref2 = _sample(F.conv_transpose3d(_densify(out, coarse, S // 2, B), wt2, stride=2), level)
    assert torch.allclose(up, ref2, atol=1e-4, rtol=1e-4)


def test_batchnorm_matches_torch_and_updates_running_stats():
    x = torch.randn(200, 9) * 3 + 1
    bn = scn.BatchNormReLU(9)
    t = scn.SparseConvNetTensor(x, scn.Level(np.zeros((0, 4), np.int64), 8), 8)
    t.root = None
    y = bn(t).features
    mean, var = x.mean(0), x.var(0, unbiased=False)
    assert torch.allclose(y, F.relu((x - mean) / torch.sqrt(var + 1e-4)), atol=1e-5)
    keep = scn.DEFAULT_BN_MOMENTUM[0]  # 0.99: the recalled constructor default of the pinned commit (the docstring says 0.9)
    assert keep == 0.99 and bn.momentum == keep
    assert torch.allclose(bn.running_mean, (1 - keep) * mean, atol=1e-6)
    assert torch.allclose(bn.running_var, keep + (1 - keep) * x.var(0, unbiased=True), atol=1e-5)
    bn9 = scn.BatchNormReLU(9, momentum=0.9)  # the survey's reading stays selectable
    bn9(t)
    assert torch.allclose(bn9.running_mean, 0.1 * mean, atol=1e-6)


def test_unet_forward_backward_and_output_order():
    from oracle.net3d_ref import Net3DSegRef

    torch.manual_seed(0)
    coords, feats = _random_sparse(5, S=64, B=2, n=400, C=3)
    net = Net3DSegRef(6, True, dict(in_channels=3, m=16, full_scale=64, num_planes=4))
    batch = {"x": [coords, feats.clone()]}
    preds, feat, aux = net(batch)
    assert preds["seg_logit"].shape == (len(coords), 6) and feat.shape == (len(coords), 16)
    # duplicate points share their voxel's row (OutputLayer copies, no division)
    keys = scn.pack_keys(coords.numpy())
    _, inv = np.unique(keys, return_inverse=True)
    a, b = np.nonzero(inv[:, None] == inv[None, :])
    assert torch.equal(feat[a], feat[b])
    (preds["seg_logit"].sum() + aux["seg_logit_point"].sum()).backward()
    assert all(p.grad is not None for n, p in net.named_parameters() if "linear_global" not in n)
