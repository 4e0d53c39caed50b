"""numpy emulation of how nwe_mfma_kernels.h (mlp_eval) consumes the packed weight stream.

It replays mlp_eval() with the documented lane maps of v_mfma_f32_32x32x16_f16 (A: lane (i,h) holds
A[i][8h+j]; B: lane (n,h) holds B[8h+j][n]; D: lane (n,h) register r holds D[(r&3)+8(r>>2)+4h][n]) on the
byte stream the C ABI produced, so the packer's permutations (gamma_col / hidden_col / chunk order /
bias tile / duplicated head rows) can be checked on the CPU against a plain matmul.  The hardware side
of the same assumptions is checked on the GPU by nwe_selftest().
"""
import numpy as np

TILE = 1024


def split(v):
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


class Stream:
    """The packed network: tile stream (hi, lo per k-step), per-chunk bias table, weight scale."""

    def __init__(self, buf: np.ndarray, bias: np.ndarray, scale: float):
        self.buf, self.bias_tab, self.scale = buf, bias, np.float32(scale)
        self.pos = 0
        self.chunk = 0

    def tile_f16(self):
        t = self.buf[self.pos:self.pos + TILE].view(np.float16).reshape(64, 8)   # [lane, j]
        self.pos += TILE
        return t

    def bias(self):
        b = self.bias_tab[self.chunk].copy()
        self.chunk += 1
        return b


def encode(v3, nb, nk):
    """v3 [3, n] -> (hi, lo) arrays [nk, 2(h), 8(j), n], the kernel's encode<NB, NK>()."""
    n = v3.shape[1]
    vals = np.zeros((2, nk * 8, n), dtype=np.float32)
    for h in range(2):
        for bl in range(nb):
            f = np.float32(2.0 ** (bl + nb * h))
            for c in range(3):
                arg = (v3[c] * f).astype(np.float32)
                vals[h, 2 * (bl * 3 + c)] = np.sin(arg.astype(np.float64)).astype(np.float32)
                vals[h, 2 * (bl * 3 + c) + 1] = np.cos(arg.astype(np.float64)).astype(np.float32)
        vals[h, 6 * nb] = v3[2] if h else v3[0]
        vals[h, 6 * nb + 1] = 0 if h else v3[1]
    hi, lo = split(vals)
    shp = lambda a: a.reshape(2, nk, 8, n).transpose(1, 0, 2, 3)
    return shp(hi), shp(lo)


def mma_tile(st: Stream, segs, three_pass=True):
    """One chunk: bias tile + k-steps.  segs = list of (Xhi, Xlo) with shape [ksteps, 2, 8, n].
    Returns the [32, n] fp32 tile (row-major, i.e. already un-permuted from the D register map)."""
    bias = st.bias()
    n = segs[0][0].shape[-1]
    acc = np.zeros((32, n), dtype=np.float64)
    for xhi, xlo in segs:
        for s in range(xhi.shape[0]):
            a_hi = st.tile_f16().astype(np.float64).reshape(2, 32, 8)   # [h, i, j]
            a_lo = st.tile_f16().astype(np.float64).reshape(2, 32, 8)
            bh = xhi[s].astype(np.float64)                               # [h, j, n]
            bl = xlo[s].astype(np.float64)
            acc += np.einsum("hij,hjn->in", a_hi, bh)
            if three_pass:
                acc += np.einsum("hij,hjn->in", a_lo, bh) + np.einsum("hij,hjn->in", a_hi, bl)
    return (acc.astype(np.float32) / st.scale + bias[:, None]).astype(np.float32)


def tile_to_operand(tile, lower):
    """[32, n] result tile -> two k-steps of the next B operand, [2, 2(h), 8(j), n] (split_tile())."""
    v = np.maximum(tile, np.float32(lower))
    n = v.shape[1]
    out = np.zeros((2, 2, 8, n), dtype=np.float32)
    for h in range(2):
        for r in range(16):
            out[r >> 3, h, r & 7] = v[(r & 3) + 8 * (r >> 2) + 4 * h]
    hi, lo = split(out)
    return hi, lo


def layer(st, n_tiles, segs, lower, three_pass=True):
    his, los = [], []
    for _ in range(n_tiles):
        t = mma_tile(st, segs, three_pass)
        hi, lo = tile_to_operand(t, lower)
        his.append(hi)
        los.append(lo)
    return np.concatenate(his, 0), np.concatenate(los, 0)


def mlp_eval(stream_bytes, bias_tab, scale, pts, dirs, D, W, skip, three_pass=True, folded=True, no_view_dirs=False):
    """pts [n,3] (already divided by 10), dirs [n,3] -> raw [n,4] as the kernel would produce.  folded: the stream has no
    feature-layer chunks (the packer multiplied _feature_linear into the view layer), the kernel's FOLD path.
    no_view_dirs (dirs unused): the trunk, then one chunk of _output_linear whose rows 0..3 are rgb_raw, sigma_raw
    (kFormNoViewDirs)."""
    st = Stream(stream_bytes, bias_tab, scale)
    G = encode(pts.T.astype(np.float32), 5, 4)
    NT = W // 32
    if no_view_dirs:
        A = layer(st, NT, [G], 0.0, three_pass)
        for li in range(1, D):                             # trunk layers 1..D-1, ReLU each (nerf_model.py:55-59)
            A = layer(st, NT, ([G] if li == skip + 1 and skip >= 0 else []) + [A], 0.0, three_pass)
        t = mma_tile(st, [A], three_pass)                  # _output_linear (:78-79)
        assert np.array_equal(t[0:4], t[4:8]), "output tile rows 4..7 must copy rows 0..3"
        assert st.pos == len(stream_bytes), (st.pos, len(stream_bytes))
        assert st.chunk == len(bias_tab), (st.chunk, len(bias_tab))
        return np.stack([t[0], t[1], t[2], t[3]], axis=1)
    GD = encode(dirs.T.astype(np.float32), 2, 2)
    A = layer(st, NT, [G], 0.0, three_pass)
    npair = D // 2
    skip_pair = -1 if skip < 0 else skip // 2
    h_tiles = []                                           # folded: the fp32 activations of the last trunk layer, tile by tile
    for pair in range(npair):
        last = pair == npair - 1
        segs = ([G] if pair == skip_pair else []) + [A]
        if folded and last:
            his, los = [], []
            for _ in range(NT):
                t = mma_tile(st, segs, three_pass)
                h_tiles.append(np.maximum(t, np.float32(0.0)))
                hi, lo = tile_to_operand(t, 0.0)
                his.append(hi); los.append(lo)
            B = (np.concatenate(his, 0), np.concatenate(los, 0))
            break
        B = layer(st, NT, segs, 0.0, three_pass)
        A = layer(st, NT, [B], -np.inf if last else 0.0, three_pass)
    if folded:
        # _alpha_linear is not a tile of the folded stream: sigma = w . h + b in fp32 on the vector ALU, accumulated with the
        # epilogues of the last trunk layer's tiles.  Its weights are the NT rows behind the chunks' bias rows (row rt element i
        # = weight of feature 32 rt + i), its bias element 0 of the row after them.  Each lane half sums its own rows
        # (8g + 4h + i of every tile, in tile / group / element order), the halves are added, then the bias.
        n_chunks = len(bias_tab) - (NT + 1)
        dot = bias_tab[n_chunks:]
        part = np.zeros((2, pts.shape[0]), dtype=np.float32)
        for rt in range(NT):
            for e in range(16):
                for hh in range(2):
                    row = 8 * (e >> 2) + 4 * hh + (e & 3)
                    part[hh] = (dot[rt, row].astype(np.float64) * h_tiles[rt][row].astype(np.float64) + part[hh].astype(np.float64)).astype(np.float32)   # fmaf
        sigma = ((part[0] + part[1]).astype(np.float32) + dot[NT, 0]).astype(np.float32)
    else:
        t = mma_tile(st, [B], three_pass)                  # _alpha_linear reads h = B, the input of _feature_linear
        assert np.array_equal(t[0], t[4]), "alpha tile rows 0 and 4 must be copies"
        sigma = t[0]
        n_chunks = len(bias_tab)
    Bv = layer(st, W // 64, [B if folded else A, GD], 0.0, three_pass)
    t = mma_tile(st, [Bv], three_pass)
    assert np.array_equal(t[0:3], t[4:7]), "rgb tile rows 4..6 must copy rows 0..2"
    assert st.pos == len(stream_bytes), (st.pos, len(stream_bytes))
    assert st.chunk == n_chunks, (st.chunk, n_chunks)
    return np.stack([t[0], t[1], t[2], sigma], axis=1)
