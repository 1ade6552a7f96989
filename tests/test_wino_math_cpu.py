"""The Winograd identities the HIP kernels implement (csrc/wino.inc), checked in numpy on the CPU:
  forward / dgrad   Y  = At [ (G g Gt) (.) (Bt d B) ] A                      F(2x2, 3x3)
  weight gradient   dg = Gt [ sum_tiles (A dY At) (.) (Bt d B) ] G           F(3x3, 2x2)
with exactly the matrices, signs and the row-a / column-b splitting used by the kernels' per-thread pieces."""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def direct(d, g):
    return np.array([[np.sum(d[i:i + 3, j:j + 3] * g) for j in range(2)] for i in range(2)])


def test_forward_identity():
    rng = np.random.default_rng(0)
    for _ in range(20):
        d, g = rng.standard_normal((4, 4)), rng.standard_normal((3, 3))
        y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
        assert np.allclose(y, direct(d, g), atol=1e-12)


def test_input_transform_pieces():
    """wino_kernel's thread computes row a of Bt d as  s0*d[r0] + s1*d[r1]  and then the four columns
    t0-t2, t1+t2, t2-t1, t1-t3."""
    rng = np.random.default_rng(1)
    d = rng.standard_normal((4, 4))
    ref = BT @ d @ BT.T
    for a in range(4):
        r0, r1 = (1 if a else 0), (3 if a == 3 else 2)
        s0, s1 = (-1.0 if a == 2 else 1.0), (-1.0 if a in (0, 3) else 1.0)
        t = s0 * d[r0] + s1 * d[r1]
        row = np.array([t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]])
        assert np.allclose(row, ref[a], atol=1e-12)


def test_output_transform_halves():
    """Epilogue of wino_kernel: wave half xh holds rows a = 2xh, 2xh+1 of M; row sums s[a][j]; the xh=0 half finishes
    output row 0 (own s0+s1, needs s2), the xh=1 half output row 1 (own -s2-s3, needs s1)."""
    rng = np.random.default_rng(2)
    m = rng.standard_normal((4, 4))
    s = np.stack([m[:, 0] + m[:, 1] + m[:, 2], m[:, 1] - m[:, 2] - m[:, 3]], axis=1)      # s[a][j]
    y0 = (s[0] + s[1]) + s[2]
    y1 = (-s[2] - s[3]) + s[1]
    assert np.allclose(np.stack([y0, y1]), AT @ m @ AT.T, atol=1e-12)


def test_wgrad_identity():
    rng = np.random.default_rng(3)
    tiles = [(rng.standard_normal((4, 4)), rng.standard_normal((2, 2))) for _ in range(7)]
    ref = sum(np.array([[np.sum(d[u:u + 2, v:v + 2] * dy) for v in range(3)] for u in range(3)]) for d, dy in tiles)
    A = AT.T
    du = sum((A @ dy @ A.T) * (BT @ d @ BT.T) for d, dy in tiles)
    assert np.allclose(G.T @ du @ G, ref, atol=1e-12)
    # the kernel's pieces: row a of A dY = c0*D0 + c1*D1, columns r0, r0+r1, r0-r1, -r1; reduce: Gt . G in two passes
    d, dy = tiles[0]
    for a in range(4):
        c0, c1 = (0.0 if a == 3 else 1.0), (0.0 if a == 0 else (1.0 if a == 1 else -1.0))
        r = c0 * dy[0] + c1 * dy[1]
        assert np.allclose(np.array([r[0], r[0] + r[1], r[0] - r[1], -r[1]]), (A @ dy @ A.T)[a], atol=1e-12)
    t = np.stack([du[0] + .5 * (du[1] + du[2]), .5 * (du[1] - du[2]), .5 * (du[1] + du[2]) + du[3]])
    dw = np.stack([t[:, 0] + .5 * (t[:, 1] + t[:, 2]), .5 * (t[:, 1] - t[:, 2]), .5 * (t[:, 1] + t[:, 2]) + t[:, 3]], axis=1)
    assert np.allclose(dw, ref, atol=1e-12)


def test_lds_row_swizzle_is_conflict_free():
    """ds_read_b128 services a wave in four 16-lane groups (MI355X_MICROARCH.md, LDS); rows are 8 floats, lane half h
    fetches k-half h, and the two 16-byte halves of a row are swapped when bit 4 of the row is set: every group must
    touch 64 distinct banks."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in g_] for g_ in groups]
    for base_row in (0, 32):
        for grp in groups:
            banks = set()
            for lane in grp:
                row = base_row + (lane & 31)
                off = row * 8 + ((((lane >> 5) ^ (row >> 4)) & 1) << 2)
                banks.update((off + e) % 64 for e in range(4))
            assert len(banks) == 64
