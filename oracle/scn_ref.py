"""CPU oracle: SparseConvNet-style operators (TEST INFRASTRUCTURE ONLY).

Restates, on torch-CPU/numpy, the operator semantics the reference's 3D branch
obtains from its un-vendored dependency ``sparseconvnet`` (pinned at
/root/reference/environment.yml:37).  Call sites restated:
  InputLayer mode 4 ............ EXP/3d_net/scn_unet.py:113,121
  SubmanifoldConvolution 3^3 ... EXP/3d_net/scn_unet.py:43,45,52,114
  Convolution k2 s2 ............ EXP/3d_net/scn_unet.py:68-70
  Deconvolution k2 s2 .......... EXP/3d_net/scn_unet.py:75-77
  BatchNorm(Leaky)ReLU ......... EXP/3d_net/scn_unet.py:42,44,51,66,73,116
  OutputLayer .................. EXP/3d_net/scn_unet.py:117,125
  Sequential/ConcatTable/JoinTable/AddTable/Identity/NetworkInNetwork
                                 EXP/3d_net/scn_unet.py:38-47,56-84
(EXP = /root/reference/experiments_USA_SING/rgbd_rgbxyz_sigmoid_for_rgb.)

Operator semantics follow SURVEY.md Appendix A; canonical orders follow A.8:
  (i)   level-0 ids      = first occurrence over the concatenated [N,4] input
  (ii)  level-(l+1) ids  = first occurrence of parent keys scanning level-l
                           sites in id order
  (iii) inside a rulebook bucket pairs are sorted by out id
  (iv)  accumulation across offsets in ascending k.
Parity status: "parity unpinned" by the reference (no golden vectors exist
for the sparseconvnet boundary); pinned by dense equivalence with torch CPU
conv3d/conv_transpose3d/batch_norm in tests/test_oracle_dense.py.

All feature math is torch CPU fp32 and differentiable through torch autograd,
which is what makes this file the backward oracle as well.
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

KEY_BITS = 16  # per coordinate; batch index takes the top 16 bits

# 16-bit activation emulation (BASELINE.json configs[4]; the reference's SparseConvNet is fp32-only, so this mode has no
# reference behaviour - it emulates WHERE the HIP path of mm2d3d_amd/scn stores 16-bit values, so that a test against it
# measures accumulation order, not the storage format): None = plain fp32; torch.bfloat16 / torch.float16 = every sparse
# row between the stem's output and the OutputLayer is rounded to that type when it is stored (conv outputs, batch-norm
# outputs; gradients on the way back through the same points), the 16-channel-multiple convolutions multiply weights
# rounded to that type (fp32 master weights keep the gradient), all sums and the batch-norm statistics stay fp32.
EMULATE16 = [None]


class _Round16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.to(dt).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(g.dtype), None


def _q(x):
    dt = EMULATE16[0]
    return x if dt is None else _Round16.apply(x, dt)


def _qw(w, wide):
    dt = EMULATE16[0]
    if dt is None or not wide:
        return w
    return w + (w.detach().to(dt).to(w.dtype) - w.detach())  # rounded value forward, identity gradient to the fp32 master


def pack_keys(coords: np.ndarray) -> np.ndarray:
    """(x, y, z, b) int64 rows -> one uint64 key per row; requires 0 <= c < 2**16."""
    c = np.asarray(coords, dtype=np.int64)
    if c.size and (c.min() < 0 or c[:, :3].max() >= (1 << KEY_BITS) or c[:, 3].max() >= (1 << KEY_BITS)):
        raise ValueError("coordinates out of the packable range [0, 65536)")
    return (
        (c[:, 3].astype(np.uint64) << np.uint64(3 * KEY_BITS))
        | (c[:, 0].astype(np.uint64) << np.uint64(2 * KEY_BITS))
        | (c[:, 1].astype(np.uint64) << np.uint64(KEY_BITS))
        | c[:, 2].astype(np.uint64)
    )


def first_occurrence_ids(keys: np.ndarray):
    """Return (ids_per_row int32 [N], first_row_of_id int64 [n_active]).

    ids are assigned in order of first occurrence (SURVEY A.1 / A.8 i, ii)."""
    if keys.size == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int64)
    uniq, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")  # unique-slot -> rank by first occurrence
    id_of_slot = np.empty(len(uniq), np.int32)
    id_of_slot[order] = np.arange(len(uniq), dtype=np.int32)
    return id_of_slot[inv.reshape(-1)], first[order]


class _Lookup:
    """key -> id lookup on sorted unique keys."""

    def __init__(self, keys: np.ndarray):
        self.order = np.argsort(keys, kind="stable")
        self.sorted = keys[self.order]

    def __call__(self, q: np.ndarray) -> np.ndarray:
        pos = np.searchsorted(self.sorted, q)
        pos_c = np.minimum(pos, len(self.sorted) - 1) if len(self.sorted) else pos
        hit = (pos < len(self.sorted)) & (self.sorted[pos_c] == q) if len(self.sorted) else np.zeros(len(q), bool)
        out = np.full(len(q), -1, np.int64)
        out[hit] = self.order[pos_c[hit]]
        return out


class Level:
    """Active set of one spatial scale: coords int64 [n,4] in id order."""

    def __init__(self, coords: np.ndarray, spatial_size: int):
        self.coords = np.ascontiguousarray(coords, dtype=np.int64)
        self.spatial_size = int(spatial_size)
        self.n = len(self.coords)
        self._lookup = None
        self.subm = None  # rulebook (K=27)
        self.down = None  # (rulebook K=8, coarse Level)

    def lookup(self):
        if self._lookup is None:
            self._lookup = _Lookup(pack_keys(self.coords))
        return self._lookup


class Rulebook:
    """K buckets of (in, out) pairs, stored k-major; pairs sorted by out inside a bucket."""

    def __init__(self, K, rin, rout, offsets):
        self.K = K
        self.rin = np.ascontiguousarray(rin, dtype=np.int32)
        self.rout = np.ascontiguousarray(rout, dtype=np.int32)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)  # [K+1]

    @property
    def n_rules(self):
        return int(self.offsets[-1])

    def bucket(self, k):
        a, b = int(self.offsets[k]), int(self.offsets[k + 1])
        return self.rin[a:b], self.rout[a:b]


def subm_rulebook(level: Level, filter_size: int = 3) -> Rulebook:
    """offset k = ((dx+1)*3 + (dy+1))*3 + (dz+1); pair iff neighbour active."""
    if level.subm is not None:
        return level.subm
    assert filter_size == 3
    c = level.coords
    look = level.lookup()
    S = level.spatial_size
    rin, rout, offs = [], [], [0]
    for k in range(27):
        d = np.array([k // 9 - 1, (k // 3) % 3 - 1, k % 3 - 1], np.int64)
        q = c.copy()
        q[:, :3] += d
        ok = np.all((q[:, :3] >= 0) & (q[:, :3] < S), axis=1)
        ids = np.full(level.n, -1, np.int64)
        if ok.any():
            ids[ok] = look(pack_keys(q[ok]))
        o = np.nonzero(ids >= 0)[0]
        rin.append(ids[o])
        rout.append(o)
        offs.append(offs[-1] + len(o))
    level.subm = Rulebook(27, np.concatenate(rin), np.concatenate(rout), np.array(offs))
    return level.subm


def down_rulebook(level: Level):
    """SURVEY A.3 + A.8(ii): parent = floor(c/2); k = ((x&1)*2 + (y&1))*2 + (z&1)."""
    if level.down is not None:
        return level.down
    c = level.coords
    pc = c.copy()
    pc[:, :3] >>= 1
    pid, first = first_occurrence_ids(pack_keys(pc))
    coarse = Level(pc[first], level.spatial_size // 2)
    kk = ((c[:, 0] & 1) * 2 + (c[:, 1] & 1)) * 2 + (c[:, 2] & 1)
    rin, rout, offs = [], [], [0]
    for k in range(8):
        sel = np.nonzero(kk == k)[0]
        o = pid[sel]
        order = np.argsort(o, kind="stable")
        rin.append(sel[order])
        rout.append(o[order])
        offs.append(offs[-1] + len(sel))
    level.down = (Rulebook(8, np.concatenate(rin), np.concatenate(rout), np.array(offs)), coarse)
    return level.down


# --------------------------------------------------------------------------- tensors / functional


class SparseConvNetTensor:
    def __init__(self, features=None, level: Optional[Level] = None, spatial_size=None, metadata=None):
        self.features = features
        self.level = level
        self.metadata = metadata if metadata is not None else level
        self.spatial_size = spatial_size

    def __repr__(self):
        return f"SparseConvNetTensor<oracle>(features={tuple(self.features.shape)}, spatial_size={self.spatial_size})"


def rule_conv(x: torch.Tensor, w: torch.Tensor, rb: Rulebook, n_out: int, transpose_roles=False):
    """out[o] += x[i] @ w[k], k ascending (SURVEY A.2/A.8 iv).  w: [K, Cin, Cout]."""
    out = torch.zeros(n_out, w.shape[2], dtype=x.dtype)
    for k in range(rb.K):
        i, o = rb.bucket(k)
        if len(i) == 0:
            continue
        if transpose_roles:
            i, o = o, i
        it = torch.from_numpy(i.astype(np.int64))
        ot = torch.from_numpy(o.astype(np.int64))
        out = out.index_add(0, ot, x.index_select(0, it) @ w[k])
    return out


def input_layer(coords: torch.Tensor, feats: torch.Tensor, spatial_size: int, mode: int = 4):
    """SURVEY A.1.  coords Long [N,4] (x,y,z,batch); returns (features, Level, point->voxel ids)."""
    c = coords.detach().cpu().numpy().astype(np.int64)
    if c.shape[1] == 3:
        c = np.concatenate([c, np.zeros((len(c), 1), np.int64)], 1)
    p2v, first = first_occurrence_ids(pack_keys(c))
    level = Level(c[first], spatial_size)
    p2v_t = torch.from_numpy(p2v.astype(np.int64))
    n = level.n
    if mode == 4 or mode == 3:
        out = torch.zeros(n, feats.shape[1], dtype=feats.dtype).index_add(0, p2v_t, feats)
        if mode == 4:
            cnt = torch.zeros(n, dtype=feats.dtype).index_add(0, p2v_t, torch.ones(len(p2v_t), dtype=feats.dtype))
            out = out / cnt[:, None]
    elif mode == 2 or mode == 0:
        out = feats.index_select(0, torch.from_numpy(first))
    elif mode == 1:
        last = np.zeros(n, np.int64)
        last[p2v] = np.arange(len(p2v))
        out = feats.index_select(0, torch.from_numpy(last))
    else:
        raise ValueError(mode)
    return out, level, p2v_t


def batchnorm_relu(x, weight, bias, running_mean, running_var, training, eps=1e-4, momentum=0.9, leakiness=0.0):
    """scn momentum 0.9 is the keep-fraction == torch momentum 0.1."""
    y = F.batch_norm(x, running_mean, running_var, weight, bias, training, 1.0 - momentum, eps)
    return F.leaky_relu(y, leakiness) if leakiness != 0 else F.relu(y)


# --------------------------------------------------------------------------- nn.Module surface (sparseconvnet API subset)


class Sequential(nn.Sequential):
    def add(self, module):
        self._modules[str(len(self._modules))] = module
        return self

    def input_spatial_size(self, out_size):
        for m in reversed(self._modules):
            out_size = self._modules[m].input_spatial_size(out_size)
        return out_size


class InputLayer(nn.Module):
    def __init__(self, dimension, spatial_size, mode=3):
        super().__init__()
        assert dimension == 3
        self.dimension = dimension
        self.spatial_size = spatial_size
        self.mode = mode

    def forward(self, x):
        coords, feats = x[0], x[1]
        f, level, p2v = input_layer(coords, feats.cpu(), int(self.spatial_size), self.mode)
        t = SparseConvNetTensor(f, level, self.spatial_size)
        level.p2v = p2v
        t.root = level
        return t


class OutputLayer(nn.Module):
    def __init__(self, dimension):
        super().__init__()

    def forward(self, x):
        return x.features.index_select(0, x.root.p2v)


def _carry(x, features, level=None):
    t = SparseConvNetTensor(features, level if level is not None else x.level, x.spatial_size)
    t.root = x.root
    return t


class SubmanifoldConvolution(nn.Module):
    def __init__(self, dimension, nIn, nOut, filter_size, bias, groups=1):
        super().__init__()
        assert dimension == 3 and filter_size == 3 and groups == 1
        self.nIn, self.nOut = nIn, nOut
        self.filter_volume = 27
        std = (2.0 / nIn / self.filter_volume) ** 0.5
        self.weight = nn.Parameter(torch.Tensor(27, 1, nIn, nOut).normal_(0, std))
        if bias:
            self.bias = nn.Parameter(torch.zeros(nOut))

    def forward(self, x):
        rb = subm_rulebook(x.level)
        wide = self.nIn % 16 == 0 and self.nOut % 16 == 0  # the 3-channel stem multiplies in fp32, its OUTPUT enters the 16-bit region
        f = rule_conv(_q(x.features) if wide else x.features, _qw(self.weight[:, 0], wide), rb, x.level.n)
        if hasattr(self, "bias"):
            f = f + self.bias
        return _carry(x, _q(f))


class Convolution(nn.Module):
    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias, groups=1):
        super().__init__()
        assert dimension == 3 and filter_size == 2 and filter_stride == 2 and groups == 1
        self.nIn, self.nOut = nIn, nOut
        std = (2.0 / nIn / 8) ** 0.5
        self.weight = nn.Parameter(torch.Tensor(8, 1, nIn, nOut).normal_(0, std))
        if bias:
            self.bias = nn.Parameter(torch.zeros(nOut))

    def forward(self, x):
        rb, coarse = down_rulebook(x.level)
        f = rule_conv(x.features, _qw(self.weight[:, 0], True), rb, coarse.n)
        if hasattr(self, "bias"):
            f = f + self.bias
        t = _carry(x, _q(f), coarse)
        coarse.parent_fine = x.level
        return t


class Deconvolution(nn.Module):
    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias, groups=1):
        super().__init__()
        assert dimension == 3 and filter_size == 2 and filter_stride == 2 and groups == 1
        self.nIn, self.nOut = nIn, nOut
        std = (2.0 / nIn / 8) ** 0.5
        self.weight = nn.Parameter(torch.Tensor(8, 1, nIn, nOut).normal_(0, std))
        if bias:
            self.bias = nn.Parameter(torch.zeros(nOut))

    def forward(self, x):
        fine = x.level.parent_fine  # reuse the Convolution rulebook, roles swapped
        rb, coarse = down_rulebook(fine)
        assert coarse is x.level
        f = rule_conv(x.features, _qw(self.weight[:, 0], True), rb, fine.n, transpose_roles=True)
        if hasattr(self, "bias"):
            f = f + self.bias
        return _carry(x, _q(f), fine)


DEFAULT_BN_MOMENTUM = [0.99]  # constructor signature of the pinned commit as recalled (SURVEY A.5 / its docstring say 0.9; see mm2d3d_amd/scn)


class BatchNormalization(nn.Module):
    def __init__(self, nPlanes, eps=1e-4, momentum=None, affine=True, leakiness=1):
        super().__init__()
        momentum = DEFAULT_BN_MOMENTUM[0] if momentum is None else momentum
        self.nPlanes, self.eps, self.momentum, self.leakiness = nPlanes, eps, momentum, leakiness
        self.register_buffer("running_mean", torch.zeros(nPlanes))
        self.register_buffer("running_var", torch.ones(nPlanes))
        if affine:
            self.weight = nn.Parameter(torch.ones(nPlanes))
            self.bias = nn.Parameter(torch.zeros(nPlanes))
        else:
            self.weight = self.bias = None

    def forward(self, x):
        y = F.batch_norm(
            x.features, self.running_mean, self.running_var, self.weight, self.bias,
            self.training, 1.0 - self.momentum, self.eps,
        )
        if self.leakiness != 1:
            y = F.leaky_relu(y, self.leakiness) if self.leakiness != 0 else F.relu(y)
        return _carry(x, _q(y))


class BatchNormReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=None):
        super().__init__(nPlanes, eps, momentum, True, 0)


class BatchNormLeakyReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=None, leakiness=0.333):
        super().__init__(nPlanes, eps, momentum, True, leakiness)


class Identity(nn.Module):
    def forward(self, x):
        return x


class ConcatTable(nn.Sequential):
    def add(self, module):
        self._modules[str(len(self._modules))] = module
        return self

    def forward(self, x):
        return [m(x) for m in self._modules.values()]


class JoinTable(nn.Module):
    def forward(self, xs):
        return _carry(xs[0], torch.cat([t.features for t in xs], 1))


class AddTable(nn.Module):
    def forward(self, xs):
        return _carry(xs[0], sum(t.features for t in xs))


class NetworkInNetwork(nn.Module):
    def __init__(self, nIn, nOut, bias):
        super().__init__()
        std = (2.0 / nIn) ** 0.5
        self.weight = nn.Parameter(torch.Tensor(nIn, nOut).normal_(0, std))
        if bias:
            self.bias = nn.Parameter(torch.zeros(nOut))

    def forward(self, x):
        f = x.features @ self.weight
        if hasattr(self, "bias"):
            f = f + self.bias
        return _carry(x, f)
