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

MATS = {
    2: dict(  # F(2x2,3x3): 4x4 patches, 16 transform points
        BT=np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64),
        G=np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64),
        AT=np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)),
    4: dict(  # F(4x4,3x3): 6x6 patches, 36 transform points (interpolation points 0, +-1, +-2, inf)
        BT=np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                     [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64),
        G=np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                    [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64),
        AT=np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]],
                    dtype=np.float64)),
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
