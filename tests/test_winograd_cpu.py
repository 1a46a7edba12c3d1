"""The transform conventions of csrc/igemm_wino.hip, restated in numpy and checked against a direct convolution in
float64: F(2x2, 3x3), Y = A^T [ (G g G^T) . (B^T d B) ] A, 'same' padding, cross-correlation (Keras Conv2D), frequency
index f = 4a + b with a along the rows (the wave that owns it), and the two halves of the output transform as the
kernel splits them (row half per wave in registers, column half in the fused epilogue).  CPU only: this pins the
algorithm the kernel and the packing kernel implement, the GPU parity tests (tests/test_gpu_ops.py, path 8) pin the
kernel."""
import numpy as np

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)          # pack_weights_batch_kernel
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)  # rows a: d0-d2, d1+d2, d2-d1, d1-d3
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)                               # even rows Z0+Z1+Z2, odd Z1-Z2-Z3


def direct(x, w):
    H, W, C = x.shape
    xp = np.zeros((H + 2, W + 2, C))
    xp[1:-1, 1:-1] = x
    out = np.zeros((H, W, w.shape[3]))
    for i in range(3):
        for j in range(3):
            out += xp[i:i + H, j:j + W].reshape(-1, C) .dot(w[i, j]).reshape(H, W, -1)
    return out


def winograd(x, w):
    H, W, C = x.shape
    K = w.shape[3]
    xp = np.zeros((H + 2, W + 2, C))
    xp[1:-1, 1:-1] = x
    U = np.einsum("ai,ijck,bj->abck", G, w, G).reshape(16, C, K)            # panel planes f = 4a + b
    th, tw = H // 2, W // 2
    d = np.stack([[xp[i:i + H:2, j:j + W:2][:th, :tw] for j in range(4)] for i in range(4)])   # (4,4,th,tw,C)
    V = np.einsum("ai,ijtuc,bj->abtuc", BT, d, BT).reshape(16, th, tw, C)
    M = np.stack([V[f].reshape(-1, C).dot(U[f]).reshape(th, tw, K) for f in range(16)]).reshape(4, 4, th, tw, K)
    # row half (wave a reduces its four b's to the output columns q), then the column half (epilogue, over a)
    Z = np.einsum("qb,abtuk->aqtuk", AT, M)
    Y = np.einsum("pa,aqtuk->pqtuk", AT, Z)
    out = np.zeros((H, W, K))
    for p in range(2):
        for q in range(2):
            out[p::2, q::2] = Y[p, q]
    return out


def test_f2x2_3x3_equals_direct_convolution():
    rng = np.random.default_rng(0)
    for H, W, C, K in ((8, 16, 8, 32), (6, 10, 24, 32), (16, 16, 40, 64)):
        x = rng.standard_normal((H, W, C))
        w = rng.standard_normal((3, 3, C, K))
        assert np.abs(winograd(x, w) - direct(x, w)).max() < 1e-11


def test_flipped_transposed_panel_is_the_backward_data_convolution():
    """The backward-data panel is the Winograd transform of the flipped, transposed kernel (dg_pack_job(transpose=1,
    flip=1)): conv(dy, flip(w)^T) = d/dx <conv(x, w), dy>."""
    rng = np.random.default_rng(1)
    H, W, C, K = 8, 16, 8, 32
    x = rng.standard_normal((H, W, C))
    w = rng.standard_normal((3, 3, C, K))
    dy = rng.standard_normal((H, W, K))
    wb = np.ascontiguousarray(w[::-1, ::-1].transpose(0, 1, 3, 2))           # taps flipped, roles swapped
    dx = winograd(dy, wb)
    eps = 1e-6
    num = np.zeros_like(x)
    for idx in [(0, 0, 0), (3, 7, 2), (7, 15, 7), (4, 0, 5)]:
        xp = x.copy(); xp[idx] += eps
        num[idx] = ((direct(xp, w) - direct(x, w)) * dy).sum() / eps
        assert abs(num[idx] - dx[idx]) < 1e-5 * max(1.0, abs(dx[idx]))
