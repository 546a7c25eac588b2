"""numpy restatement of the dropout of the continuation-value network.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package never does.

What the reference does (paths relative to /root/reference): `SingleLSMNet` puts `nn.Dropout(dropout)` behind
every hidden ReLU (options_model_3/options_model_3.py:85-103), trains with it (:576-586, `net.train()` is the
module default) and -- because `.eval()` is never called -- keeps it switched on in pass 2 (:637-640).
`nn.Dropout(p)` in training mode is INVERTED dropout: every activation is kept with probability 1 - p and a kept
activation is multiplied by 1 / (1 - p); the backward pass multiplies the incoming gradient by the same 0 or
1 / (1 - p).

Which units are dropped comes, in the reference, from torch's global generator after `torch.manual_seed(child)`
(:455).  That stream is torch's implementation detail and -- like numpy's ziggurat normals -- is not reproduced by
a counter-based generator; the build DEFINES its masks from Philox4x32-10 instead (one block per (row, layer,
half tile), stretched by a multiply-with-carry stream to 16 bits per unit).  This module restates that definition
bit for bit, so that everything downstream of "which units are kept" -- the masked forward pass, the gradients
through the masks, the dropout-on exercise decisions -- can be compared with the float32 / float64 restatement of
the reference's arithmetic UNDER IDENTICAL MASKS.

Pinning: the Philox core against the Random123 known-answer vectors (tests/test_oracle_golden.py), the
multiply-with-carry stretch and the unit order against the device (`omc_mlp_dropout_masks`, tests/test_gpu_dropout.py),
the Bernoulli rate and independence statistically (tests/test_dropout_oracle_cpu.py).

Definition (the build's; csrc/omc_mlp.hip `relu_dropout`, `relu_dropout_1`, and the call sites of the four kernels):

* keep16 = round((1 - p) * 65536); a unit is kept iff its 16 random bits are < keep16; a kept activation is
  multiplied by inv_keep = float32(65536 / keep16)  (p = 0.1: keep16 = 58982, keep probability 0.899994).
* A 32-unit MFMA tile is held by a wave as 16 registers in each of its two half-waves h: register r of half-wave h is
  tile slot  rho(r, h) = 8 * (r >> 2) + 4 * h + (r & 3).
* PAIR generator (two tiles per Philox block): block = Philox(ctr = (row, step, tag, 0x4d4c5031), key = seed);
  stream A starts at (x << 32) | (y | 1), stream B at (z << 32) | (w | 1); one advance is
  s <- 4294957665 * lo32(s) + hi32(s) and yields lo32(s): its low 16 bits decide register 2i, its high 16 bits
  register 2i + 1 (i = 0..7: eight advances per tile and half-wave).  Stream A serves the first tile of the pair,
  stream B the second.
* SINGLE generator (one tile per Philox block; the one-tile-per-workgroup trainer): ctr = (row, step, tag, 0x4d4c5134),
  one stream started at ((x ^ z) << 32) | ((y ^ w) | 1), same advance.
* OCTET generator (the 16-row-tile trainer, `relu_dropout_q16`): no stretching -- one Philox block per eight consecutive
  hidden units, ctr = (row, step, 0x100 * (layer + 1) + unit // 8, 0x4d4c5138); unit u takes half (u & 1) of word
  (u & 7) >> 1 (low half first).
* tag, row, step and the slot -> hidden-unit map per kernel: see `TRAIN_VARIANTS` and `apply_masks`.
"""
from __future__ import annotations

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)
MWC_A = np.uint64(4294957665)
PAIR_WORD = 0x4D4C5031
SINGLE_WORD = 0x4D4C5134

# trainer kernels (csrc/omc_mlp.hip mlp_train_kernel_choice): the library picks one per (hidden, batch)
GROUP, TILE, QUAD, Q16 = 1, 2, 3, 4
TRAIN_VARIANTS = {GROUP: "mlp_train_kernel<L> (64 units, > 32 tiles)",
                  TILE: "mlp_train_tile_kernel<H,L> (one tile per wave)",
                  QUAD: "mlp_train_quad_kernel<H,L> (one 32-row tile per workgroup)",
                  Q16: "mlp_train_q16_kernel<H,L> (one 16-row tile per workgroup: the reference's minibatch of 256 rows)"}
Q16_WORD = 0x4D4C5138


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 on arrays of counters (Random123; csrc/omc_device.h philox4x32_10) -> four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, np.uint64) & _M32 for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0 = np.uint64(int(k0) & 0xFFFFFFFF)
    k1 = np.uint64(int(k1) & 0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        c1 = p1 & _M32
        c3 = p0 & _M32
        c0, c2 = n0, n2
        k0 = (k0 + np.uint64(0x9E3779B9)) & _M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & _M32
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def keep16_of(p_drop: float) -> int:
    return int(round((1.0 - float(p_drop)) * 65536.0)) if p_drop > 0.0 else 65536


def inv_keep_of(p_drop: float) -> np.float32:
    k = keep16_of(p_drop)
    return np.float32(1.0) if k >= 65536 else np.float32(65536.0 / k)


def _slot(r, h):
    return 8 * (r >> 2) + 4 * h + (r & 3)


def _stream_bits(state):
    """Eight advances of the multiply-with-carry stream(s) -> uint32 [..., 16]: the 16 random bits of registers 0..15."""
    out = np.empty(state.shape + (16,), np.uint32)
    s = state.copy()
    for i in range(8):
        s = MWC_A * (s & _M32) + (s >> np.uint64(32))
        w = (s & _M32).astype(np.uint32)
        out[..., 2 * i] = w & np.uint32(0xFFFF)
        out[..., 2 * i + 1] = w >> np.uint32(16)
    return out


def _pair_bits(row, step, tag, seed):
    """PAIR generator -> uint32 [n, 2 tiles, 16 registers] for one half-wave (the tag carries h)."""
    x, y, z, w = philox4x32_10(row, step, tag, PAIR_WORD, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    a = (x.astype(np.uint64) << np.uint64(32)) | (y | np.uint32(1)).astype(np.uint64)
    b = (z.astype(np.uint64) << np.uint64(32)) | (w | np.uint32(1)).astype(np.uint64)
    return np.stack([_stream_bits(a), _stream_bits(b)], axis=1)


def _single_bits(row, step, tag, seed):
    """SINGLE generator -> uint32 [n, 16 registers] for one half-wave."""
    x, y, z, w = philox4x32_10(row, step, tag, SINGLE_WORD, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    s = ((x ^ z).astype(np.uint64) << np.uint64(32)) | ((y ^ w) | np.uint32(1)).astype(np.uint64)
    return _stream_bits(s)


def _tiles_to_units(bits_by_tile_slot, hidden, unit_of):
    """bits [n, hidden/32 tiles, 32 slots] -> [n, hidden] by hidden unit, unit_of(tile, slot) -> unit."""
    n = bits_by_tile_slot.shape[0]
    out = np.empty((n, hidden), np.uint32)
    for tile in range(hidden // 32):
        for s in range(32):
            out[:, unit_of(tile, s)] = bits_by_tile_slot[:, tile, s]
    return out


def _layer_bits_pairs(row, step, base_tag, seed, hidden):
    """Every 32-unit tile of one layer from the PAIR generator -> uint32 [n, tiles, 32 slots].  Pair p (tiles p, p + 1)
    uses tag = base_tag + 0x1000 * p + h."""
    n = np.asarray(row).shape[0]
    nt = hidden // 32
    out = np.empty((n, nt, 32), np.uint32)
    for p in range(0, nt, 2):
        for h in (0, 1):
            bits = _pair_bits(row, step, base_tag + 0x1000 * p + h, seed)  # [n, 2, 16]
            for r in range(16):
                out[:, p, _slot(r, h)] = bits[:, 0, r]
                out[:, p + 1, _slot(r, h)] = bits[:, 1, r]
    return out


def train_bits(variant, hidden, layers, rows, step, seed):
    """The 16 random bits of every hidden activation of one optimizer step -> uint32 [layers, n, hidden].

    rows: the dropout key of each row of the minibatch = its position in the (global) minibatch, 0-based
    (`drow` in the kernels; sharded training hands these positions over as `drop_pos`).
    step: the optimizer step, counted from 1 over the whole training run (`a.step`).
    variant: GROUP / TILE / QUAD -- which trainer kernel the library runs for this (hidden, batch)."""
    rows = np.asarray(rows, np.uint32)
    nt = hidden // 32
    out = np.empty((layers, rows.shape[0], hidden), np.uint32)
    for j in range(layers):
        base = 0x100 * (j + 1)
        if variant == GROUP:      # slot s of tile mt IS unit 32 mt + s   (unit_of(mt, r, h))
            assert hidden == 64
            b = _layer_bits_pairs(rows, step, base, seed, hidden)
            out[j] = _tiles_to_units(b, hidden, lambda tile, s: 32 * tile + s)
        elif variant == TILE:     # units dealt round robin: slot 32 mt + rho <-> unit NT rho + mt
            b = _layer_bits_pairs(rows, step, base, seed, hidden)
            out[j] = _tiles_to_units(b, hidden, lambda tile, s: nt * s + tile)
        elif variant == QUAD:     # wave w owns units 32 w .. 32 w + 31; SINGLE generator, tag + 0x10 w + h
            b = np.empty((rows.shape[0], nt, 32), np.uint32)
            for w in range(nt):
                for h in (0, 1):
                    bits = _single_bits(rows, step, base + 0x10 * w + h, seed)
                    for r in range(16):
                        b[:, w, _slot(r, h)] = bits[:, r]
            out[j] = _tiles_to_units(b, hidden, lambda tile, s: 32 * tile + s)
        elif variant == Q16:      # one Philox block per eight consecutive units, 16 bits each, no stretching
            for oct_ in range(hidden // 8):
                wd = philox4x32_10(rows, step, base + oct_, Q16_WORD, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
                for e in range(8):
                    out[j, :, 8 * oct_ + e] = (wd[e >> 1] >> np.uint32(16 * (e & 1))) & np.uint32(0xFFFF)
        else:
            raise ValueError("variant")
    return out


def train_masks(variant, hidden, layers, rows, step, seed, p_drop):
    """-> bool [layers, n, hidden]: True = the unit is kept."""
    k = keep16_of(p_drop)
    if k >= 65536:
        return np.ones((layers, len(np.asarray(rows)), hidden), bool)
    return train_bits(variant, hidden, layers, rows, step, seed) < np.uint32(k)


def apply_bits(hidden, layers, cols, t, seed):
    """Pass 2 (mlp_apply_kernel): the bits of every hidden activation of the forward pass for path columns `cols` at time
    step t -> uint32 [layers, n, hidden].  cols: the column of the path in the UNSHARDED matrix (`pk`)."""
    cols = np.asarray(cols, np.int64)
    key = (cols & 0xFFFFFFFF).astype(np.uint32)
    hi = (cols >> 32).astype(np.uint32)
    if hi.any():
        raise NotImplementedError("columns beyond 2^32 carry 2 * (col >> 32) in the tag; not needed by any test")
    out = np.empty((layers, cols.shape[0], hidden), np.uint32)
    for j in range(layers):
        base = 0x300 if j == 0 else 0x400 + 0x100 * (j - 1)
        if hidden == 32:  # one 32-unit tile has no partner: the SINGLE generator (relu_dropout_n<1> = relu_dropout_1), tag base + h
            b = np.empty((cols.shape[0], 1, 32), np.uint32)
            for h in (0, 1):
                bits = _single_bits(key, int(t), base + h, seed)
                for r in range(16):
                    b[:, 0, _slot(r, h)] = bits[:, r]
        else:
            b = _layer_bits_pairs(key, int(t), base, seed, hidden)
        out[j] = _tiles_to_units(b, hidden, lambda tile, s: 32 * tile + s)
    return out


def apply_masks(hidden, layers, cols, t, seed, p_drop):
    k = keep16_of(p_drop)
    if k >= 65536:
        return np.ones((layers, len(np.asarray(cols)), hidden), bool)
    return apply_bits(hidden, layers, cols, t, seed) < np.uint32(k)


def mlp_forward_masked(state, x, masks, p_drop):
    """SingleLSMNet in TRAINING mode (options_model_3.py:85-103 with nn.Dropout active, as at :637-640) under the given
    keep masks, float32: h = relu(W h + b) * mask / keep after every hidden layer."""
    h = np.asarray(x, np.float32)
    idx = sorted({int(k.split(".")[1]) for k in state if k.endswith("weight")})
    inv = inv_keep_of(p_drop)
    for n, li in enumerate(idx):
        W = np.asarray(state[f"net.{li}.weight"], np.float32)
        b = np.asarray(state[f"net.{li}.bias"], np.float32)
        h = h @ W.T + b
        if n + 1 < len(idx):
            h = np.maximum(h, 0) * masks[n].astype(np.float32) * inv
    return h
