"""The Winograd identities the HIP kernels are built on, checked in float64 on the CPU with the transform matrices
exactly as the kernel headers state them (csrc/ct_wino.hip, ct_wino4.hip, ct_wino_wgrad.hip, ct_wino4_wgrad.hip):

  forward           Y  = A^T [ sum_c (G g G^T) .* (B^T d B) ] A                       (= conv2d, 3x3, pad 1)
  data gradient     the same kernel on dY with channels swapped and taps rotated by 180 degrees
  weight gradient   dg = G^T [ sum_tiles (A e A^T) .* (B^T d B) ] G

No GPU: this pins the algebra (matrices, tap rotation, tile / patch geometry), the GPU tests pin the kernels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

_P, _Q = 0.75, 1.5
_P2, _Q2 = _P * _P, _Q * _Q
_N0, _NP, _NQ = _P2 * _Q2, 2 * _P2 * (_P2 - _Q2), 2 * _Q2 * (_Q2 - _P2)
MATS = {
    2: dict(  # F(2x2,3x3): 4x4 patches, 16 transform points
        BT=np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64),
        G=np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64),
        AT=np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)),
    4: dict(  # F(4x4,3x3): 6x6 patches, 36 transform points; interpolation points 0, +-3/4, +-3/2, inf, the matrices
        # as csrc/ct_wino4_points.h states them (p = 3/4, q = 3/2; rows of B^T = prod_{l != j}(x - p_l),
        # G[j] = [1, p_j, p_j^2] / N_j)
        BT=np.array([[_P2 * _Q2, 0, -(_P2 + _Q2), 0, 1, 0], [0, -_P * _Q2, -_Q2, _P, 1, 0], [0, _P * _Q2, -_Q2, -_P, 1, 0],
                     [0, -_Q * _P2, -_P2, _Q, 1, 0], [0, _Q * _P2, -_P2, -_Q, 1, 0], [0, _P2 * _Q2, 0, -(_P2 + _Q2), 0, 1]],
                    dtype=np.float64),
        G=np.array([[1 / _N0, 0, 0], [1 / _NP, _P / _NP, _P2 / _NP], [1 / _NP, -_P / _NP, _P2 / _NP],
                    [1 / _NQ, _Q / _NQ, _Q2 / _NQ], [1 / _NQ, -_Q / _NQ, _Q2 / _NQ], [0, 0, 1]], dtype=np.float64),
        AT=np.array([[1, 1, 1, 1, 1, 0], [0, _P, -_P, _Q, -_Q, 0], [0, _P2, _P2, _Q2, _Q2, 0],
                     [0, _P ** 3, -_P ** 3, _Q ** 3, -_Q ** 3, 1]], dtype=np.float64)),
}


def _patches(x, m):
    """x [N,C,H,W] with H, W multiples of m -> [N,C,TY,TX,m+2,m+2] patches of the zero-padded input, stride m."""
    xp = F.pad(x, (1, 1, 1, 1))
    return xp.unfold(2, m + 2, m).unfold(3, m + 2, m)


def _wino_forward(x, w, m):
    M = MATS[m]
    BT, G, AT = (torch.tensor(M[k]) for k in ('BT', 'G', 'AT'))
    V = torch.einsum('ij,nctxjk,lk->nctxil', BT, _patches(x, m), BT)
    U = torch.einsum('ij,kcjl,ml->kcim', G, w, G)
    Mm = torch.einsum('kcim,nctxim->nktxim', U, V)
    Y = torch.einsum('ij,nktxjl,ml->nktxim', AT, Mm, AT)
    N, K = x.shape[0], w.shape[0]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, x.shape[2], x.shape[3])


@pytest.mark.parametrize('m', [2, 4])
def test_forward_identity(m):
    g = torch.Generator().manual_seed(m)
    x = torch.randn(2, 5, 4 * m, 3 * m, generator=g, dtype=torch.float64)
    w = torch.randn(7, 5, 3, 3, generator=g, dtype=torch.float64)
    assert torch.allclose(_wino_forward(x, w, m), F.conv2d(x, w, None, 1, 1), atol=1e-11)


@pytest.mark.parametrize('m', [2, 4])
def test_data_gradient_is_the_forward_kernel_on_rotated_swapped_weights(m):
    """ct_conv_pack_weights_wino[4]_dgrad: this convolution's (co, ci) = forward (ci, co), taps [2-i][2-j]."""
    g = torch.Generator().manual_seed(10 + m)
    x = torch.zeros(2, 5, 2 * m, 2 * m, dtype=torch.float64, requires_grad=True)
    w = torch.randn(7, 5, 3, 3, generator=g, dtype=torch.float64)
    dy = torch.randn(2, 7, 2 * m, 2 * m, generator=g, dtype=torch.float64)
    F.conv2d(x, w, None, 1, 1).backward(dy)
    wd = w.flip(2, 3).transpose(0, 1).contiguous()
    assert torch.allclose(_wino_forward(dy, wd, m), x.grad, atol=1e-11)


@pytest.mark.parametrize('m', [2, 4])
def test_weight_gradient_identity(m):
    M = MATS[m]
    BT, G, AT = (torch.tensor(M[k]) for k in ('BT', 'G', 'AT'))
    g = torch.Generator().manual_seed(20 + m)
    x = torch.randn(2, 5, 3 * m, 2 * m, generator=g, dtype=torch.float64)
    w = torch.zeros(7, 5, 3, 3, dtype=torch.float64, requires_grad=True)
    dz = torch.randn(2, 7, 3 * m, 2 * m, generator=g, dtype=torch.float64)
    F.conv2d(x, w, None, 1, 1).backward(dz)
    V = torch.einsum('ij,nctxjk,lk->nctxil', BT, _patches(x, m), BT)             # B^T d B
    e = dz.unfold(2, m, m).unfold(3, m, m)                                          # [N,K,TY,TX,m,m] tiles of dZ
    E = torch.einsum('ji,nktxjl,lm->nktxim', AT, e, AT)                             # A e A^T  (A = (A^T)^T)
    dU = torch.einsum('nktxim,nctxim->kcim', E, V)
    dg = torch.einsum('ia,kcij,jb->kcab', G, dU, G)                                 # G^T dU G
    assert torch.allclose(dg, w.grad, atol=1e-10)


def test_multiplication_counts():
    """Per output pixel and (cin, cout) pair: 9 direct, 16/4 for F(2x2,3x3), 36/16 for F(4x4,3x3) -- the ratios
    bench.py uses for the executed-flop roofline (WINOGRAD_MULT_RATIO)."""
    for m, ratio in ((2, 16 / 36), (4, 36 / 144)):
        pts = MATS[m]['BT'].shape[0] ** 2
        assert pts / (m * m) / 9 == pytest.approx(ratio)


def test_f4_transform_header_matches_these_matrices():
    """The constants of csrc/ct_wino4_points.h (the one place the F(4x4,3x3) kernels take their transforms from)."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'context-transformer_amd', 'csrc',
                            'ct_wino4_points.h')).read()
    m = re.search(r'constexpr float P = ([0-9.]+)f, Q = ([0-9.]+)f;', src)
    assert m and (float(m.group(1)), float(m.group(2))) == (_P, _Q)
    # 1-D forms of the header, restated: bt6 / at4 / a6 / gmul6 / gt3 against the matrices
    rng = np.random.RandomState(0)
    d = rng.randn(6)
    a = d[4] - _Q2 * d[2]; b = _P * (d[3] - _Q2 * d[1]); c = d[4] - _P2 * d[2]; e = _Q * (d[3] - _P2 * d[1])
    bt = np.array([_P2 * _Q2 * d[0] - (_P2 + _Q2) * d[2] + d[4], a + b, a - b, c + e, c - e,
                   _P2 * _Q2 * d[1] - (_P2 + _Q2) * d[3] + d[5]])
    assert np.allclose(bt, MATS[4]['BT'] @ d)
    p, n, r, s = d[1] + d[2], d[1] - d[2], d[3] + d[4], d[3] - d[4]
    at = np.array([d[0] + p + r, _Q * s + _P * n, _Q2 * r + _P2 * p, _Q ** 3 * s + _P ** 3 * n + d[5]])
    assert np.allclose(at, MATS[4]['AT'] @ d)
    e4 = rng.randn(4)
    ep, op = e4[0] + _P2 * e4[2], _P * e4[1] + _P ** 3 * e4[3]
    eq, oq = e4[0] + _Q2 * e4[2], _Q * e4[1] + _Q ** 3 * e4[3]
    assert np.allclose(np.array([e4[0], ep + op, ep - op, eq + oq, eq - oq, e4[3]]), MATS[4]['AT'].T @ e4)
    g3 = rng.randn(3)
    gp, gop = (g3[0] + _P2 * g3[2]) / _NP, _P * g3[1] / _NP
    gq, goq = (g3[0] + _Q2 * g3[2]) / _NQ, _Q * g3[1] / _NQ
    assert np.allclose(np.array([g3[0] / _N0, gp + gop, gp - gop, gq + goq, gq - goq, g3[2]]), MATS[4]['G'] @ g3)
    sp, dp, sq, dq = d[1] + d[2], d[1] - d[2], d[3] + d[4], d[3] - d[4]
    gt = np.array([d[0] / _N0 + sp / _NP + sq / _NQ, _P / _NP * dp + _Q / _NQ * dq, _P2 / _NP * sp + _Q2 / _NQ * sq + d[5]])
    assert np.allclose(gt, MATS[4]['G'].T @ d)


@pytest.mark.parametrize('geom', [(19, 19, 6), (19, 19, 2), (38, 37, 3), (38, 38, 5), (32, 32, 6), (5, 4, 3), (1, 1, 2), (23, 9, 8)])
def test_dilated_layer_is_d2_pad1_layers_on_its_sublattices_and_the_tile_numbering(geom):
    """csrc/ct_wino4s.hip (wino4s_in_dil, the dilated branch of wino4s_out, wino4s_tk): a 3x3 convolution with dilation d and
    pad d is, on each of the d x d residue classes (sy, sx) of the pixel grid, an ordinary pad-1 3x3 convolution of the
    ceil(H / d) x ceil(W / d) sub-image -- checked against conv2d in float64 -- and the kernels' tile numbering
    T = (((n d + sy) d + sx) TY + ty) TX + tx with patch element (i, j) = pixel (sy + d (4 ty - 1 + i), sx + d (4 tx - 1 + j))
    visits every output pixel exactly once and every patch element at the pixel the sub-lattice convolution reads."""
    H, W, d = geom
    g = torch.Generator().manual_seed(H * 100 + W + d)
    N, C, K = 2, 3, 4
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(K, C, 3, 3, generator=g, dtype=torch.float64)
    want = F.conv2d(x, w, None, 1, d, d)
    got = torch.zeros_like(want)
    for sy in range(d):
        for sx in range(d):
            sub = x[:, :, sy::d, sx::d]
            if sub.numel():
                got[:, :, sy::d, sx::d] = F.conv2d(sub, w, None, 1, 1, 1)
    assert torch.allclose(got, want, atol=1e-12)
    # tile geometry as the kernels decode it
    TY, TX = (-(-H // d) + 3) // 4, (-(-W // d) + 3) // 4
    NT = N * d * d * TY * TX
    seen = torch.zeros(N, H, W, dtype=torch.int64)
    xp = F.pad(x, (d, d, d, d))
    for T in range(NT):
        q, rem = divmod(T, TY * TX)
        ty, tx = divmod(rem, TX)
        q, sx = divmod(q, d)
        n, sy = divmod(q, d)
        for r in range(4):
            for c in range(4):
                yy, xx = sy + d * (4 * ty + r), sx + d * (4 * tx + c)
                if yy < H and xx < W:
                    seen[n, yy, xx] += 1
                    # the output's 3x3 window = patch elements (r .. r + 2, c .. c + 2) of this tile
                    win = torch.stack([torch.stack([xp[n, :, d + sy + d * (4 * ty - 1 + r + a), d + sx + d * (4 * tx - 1 + c + b)]
                                                    for b in range(3)], -1) for a in range(3)], -2)      # [C, 3, 3]
                    if T % 7 == 0 and r == c:                  # a sample is enough: the loop is pure Python
                        assert torch.allclose((win.unsqueeze(0) * w).sum((1, 2, 3)), want[n, :, yy, xx], atol=1e-12)
    assert int(seen.min()) == 1 and int(seen.max()) == 1
