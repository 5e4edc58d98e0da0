"""Host-side model of the work decomposition of ``attn_prefill_lds_kernel`` (csrc/kernels/attention.hip).

The kernel cuts a ragged batch into workgroups of four waves (one 32-row query tile x four query
heads of one kv head) and walks the key tiles of two query tiles (j and nq-1-j) through a ring of
LDS stages. Everything that decides WHICH
memory a lane touches is integer arithmetic on the launch arguments, so it is checked here, lane by
lane, without a GPU: every DMA source lies inside the block the block table names and inside the
cache, every LDS destination inside its stage, the pieces of a tile are issued exactly once, every
visible (row, key) pair is processed, and every output row of every head is written exactly once.
The constants and formulas mirror the kernel; a change there that is not made here fails the
cross-check against the kernel source at the bottom of this file.
"""
import re
from pathlib import Path

import numpy as np
import pytest

PF_NSTAGE, PF_WAVES = 4, 4
SRC = Path(__file__).resolve().parents[1] / "csrc" / "kernels" / "attention.hip"


def xcd_group_remap(i, gx, n_groups):
    full = (n_groups & ~7) * gx
    if i >= full:
        return i
    x, k = i & 7, i >> 3
    return (x + 8 * (k // gx)) * gx + k % gx


def run_model(q_lens, ctx_lens, Hq, Hkv, DH, BS, window, rng):
    n_seqs = len(q_lens)
    G = Hq // Hkv
    nsub = (G + 3) // 4
    lbs = BS.bit_length() - 1
    max_q_len = max(q_lens)
    max_blocks = max((c + BS - 1) // BS for c in ctx_lens) + 1
    num_blocks = sum((c + BS - 1) // BS for c in ctx_lens) + 3
    # block tables: distinct physical blocks in a shuffled order; unused entries hold a poison value
    perm = list(rng.permutation(num_blocks))
    POISON = 10 ** 9
    bt = np.full((n_seqs, max_blocks), POISON, dtype=np.int64)
    for s, c in enumerate(ctx_lens):
        for b in range((c + BS - 1) // BS):
            bt[s, b] = perm.pop()
    cu_q = np.concatenate([[0], np.cumsum(q_lens)])
    kblk_stride = Hkv * BS * DH
    cache_elems = num_blocks * kblk_stride

    CH = DH // 8
    KEYS_PER_PIECE = 64 // CH
    K_PIECES = 32 // KEYS_PER_PIECE
    V_PIECES = DH // 16
    PIECES = K_PIECES + V_PIECES
    K_BYTES = 32 * DH * 2
    STAGE = 2 * K_BYTES
    assert PIECES % PF_WAVES == 0
    PPW = PIECES // PF_WAVES

    # host side of lc_attn_prefill
    gx, gy = ((max_q_len + 31) // 32 + 1) // 2, Hkv * nsub
    grid = gx * gy * n_seqs

    logicals = sorted(xcd_group_remap(i, gx, grid // gx) for i in range(grid))
    assert logicals == list(range(grid)), "workgroup remap is not a permutation"

    written = np.zeros((int(cu_q[-1]), Hq), dtype=np.int32)
    lanes = np.arange(64)
    r_l, h_l = lanes & 31, lanes >> 5

    for logical in range(grid):
        bx, by, s = logical % gx, (logical // gx) % gy, logical // (gx * gy)
        kvh, sub = by // nsub, by % nsub
        q0, q_len = int(cu_q[s]), q_lens[s]
        nq = (q_len + 31) >> 5
        t_heavy = nq - 1 - bx
        if t_heavy < bx:
            continue
        ctx = ctx_lens[s]
        base_pos = ctx - q_len
        assert base_pos >= 0
        kv_top = base_pos + min(t_heavy * 32 + 31, q_len - 1) + 1
        bt_loaded = ((kv_top - 1) >> lbs) + 1        # entries 0 .. bt_loaded-1 are in LDS
        assert bt_loaded <= max_blocks
        two = t_heavy != bx

        kv_his, t_los, nts = [0, 0], [0, 0], [0, 0]
        for p in range(2):
            tile = t_heavy if p == 0 else bx
            kv_his[p] = base_pos + min(tile * 32 + 31, q_len - 1) + 1
            kv_lo = 0
            if window > 0:
                first_pos = base_pos + tile * 32
                kv_lo = first_pos + 1 - window if first_pos + 1 > window else 0
            t_los[p] = kv_lo >> 5
            nts[p] = (((kv_his[p] + 31) >> 5) - t_los[p]) if (p == 0 or two) else 0
            assert nts[p] >= 0
        nt0, total = nts[0], nts[0] + nts[1]

        # ---- DMA: every virtual tile, every wave, every lane ----
        for v in range(total):
            p = 1 if v >= nt0 else 0
            kv_hi = kv_his[p]
            k0 = (t_los[p] + (v - nt0 if p else v)) << 5
            assert k0 < kv_hi, "a tile with no key inside the sequence would be fetched"
            pieces_seen = []
            for w in range(PF_WAVES):
                for q in range(PPW):
                    piece = q * PF_WAVES + w
                    pieces_seen.append(piece)
                    if piece < K_PIECES:
                        kl = piece * KEYS_PER_PIECE + lanes // CH
                        lc = lanes % CH
                        key = np.minimum(k0 + kl, kv_hi - 1)
                        assert ((key >> lbs) < bt_loaded).all()
                        blk = bt[s, key >> lbs]
                        assert (blk != POISON).all(), "K tile reads a block-table entry past the sequence"
                        src = blk * kblk_stride + (kvh * BS + (key & (BS - 1))) * DH + ((lc ^ (kl & (CH - 1))) << 3)
                        dst = piece * 1024 + lanes * 16
                        assert dst.max() + 16 <= K_BYTES
                    else:
                        vp = piece - K_PIECES
                        d = vp * 16 + (lanes >> 2)
                        lc = lanes & 3
                        key8 = k0 + 8 * (lc ^ ((d >> 2) & 3))
                        last8 = (kv_hi - 1) & ~7
                        key8 = np.where(key8 < kv_hi, key8, last8)
                        assert ((key8 >> lbs) < bt_loaded).all()
                        blk = bt[s, key8 >> lbs]
                        assert (blk != POISON).all(), "V tile reads a block-table entry past the sequence"
                        assert ((key8 & (BS - 1)) + 8 <= BS).all()
                        src = blk * kblk_stride + (kvh * DH + d) * BS + (key8 & (BS - 1))
                        dst = K_BYTES + vp * 1024 + lanes * 16
                        assert dst.max() + 16 <= STAGE
                    assert src.min() >= 0 and src.max() + 8 <= cache_elems, "DMA source outside the cache"
            assert sorted(pieces_seen) == list(range(PIECES)), "pieces of a tile not issued exactly once"

        # ---- compute: rows, visibility, output rows ----
        for w in range(PF_WAVES):
            g = w
            has_head = sub * 4 + g < G
            head = kvh * G + min(sub * 4 + g, G - 1)
            for p in range(2):
                if p == 1 and not two:
                    continue
                tile = t_heavy if p == 0 else bx
                row = tile * 32 + r_l
                valid = has_head & (row < q_len)
                pos = base_pos + row
                hi_q = np.where(valid, pos + 1, 0)
                lo_q = np.where(valid & (window > 0) & (pos + 1 > window), pos + 1 - window, 0)
                done = [(t_los[p] + t) << 5 for t in range(nts[p])]
                # fragment reads stay inside the stage
                pi_r = (r_l & ~0xC) | ((r_l & 4) << 1) | ((r_l & 8) >> 1)
                for kk in range(DH // 16):
                    off = pi_r * (DH * 2) + (((2 * kk + h_l) ^ (pi_r & (CH - 1))) << 4)
                    assert off.max() + 16 <= K_BYTES
                for sx in range(2):
                    for b in range(DH // 32):
                        d = b * 32 + r_l
                        off = K_BYTES + d * 64 + (((2 * sx + h_l) ^ ((d >> 2) & 3)) << 4)
                        assert off.max() + 16 <= STAGE
                for ln in range(32):
                    if not valid[ln]:
                        continue
                    lo, hi = int(lo_q[ln]), int(hi_q[ln])
                    assert 0 <= lo < hi <= ctx
                    first_tile, last_tile = lo >> 5, (hi - 1) >> 5
                    need = {t << 5 for t in range(first_tile, last_tile + 1)}
                    assert need <= set(done), f"row {int(row[ln])} of seq {s} misses key tiles {sorted(need - set(done))}"
                    written[q0 + int(row[ln]), head] += 1

    assert (written == 1).all(), "an output row is written twice or not at all"
    return grid


CASES = [
    # q_lens, extra context in front (prefix-cache hits / later chunks), Hq, Hkv, DH, BS, window
    ([512] * 4, [0] * 4, 32, 8, 128, 16, 0),
    ([1, 31, 32, 33, 63, 64, 65, 97, 129, 200], [0, 5, 0, 16, 100, 0, 33, 7, 64, 1], 8, 2, 128, 16, 0),
    ([300, 17, 450], [0, 40, 123], 28, 4, 128, 32, 0),      # G = 7: two head groups, one with an idle wave
    ([257, 64], [31, 0], 4, 4, 64, 8, 0),                   # G = 1
    ([700, 90], [300, 1000], 8, 2, 128, 16, 256),            # sliding window
    ([129], [4000], 16, 2, 64, 64, 1024),
    ([33] * 20, [0] * 20, 8, 8, 128, 16, 0),                 # more than 8 groups: the XCD remap permutes
]


@pytest.mark.parametrize("case", CASES)
def test_prefill_attention_index_model(case):
    q_lens, extra, Hq, Hkv, DH, BS, window = case
    rng = np.random.default_rng(0)
    ctx = [q + e for q, e in zip(q_lens, extra)]
    assert run_model(q_lens, ctx, Hq, Hkv, DH, BS, window, rng) > 0


def test_model_constants_match_kernel_source():
    src = SRC.read_text()
    assert re.search(r"constexpr int PF_NSTAGE = %d;" % PF_NSTAGE, src)
    assert "__launch_bounds__(256, 2) void attn_prefill_lds_kernel" in src   # PF_WAVES = 4 waves
    # the formulas the model mirrors, verbatim in the kernel
    for line in (
        "const int nq = (q_len + 31) >> 5;",
        "const int t_heavy = nq - 1 - bx;",
        "const int kv_top = base_pos + min(t_heavy * 32 + 31, q_len - 1) + 1;",
        "const int tile = p == 0 ? t_heavy : bx;",
        "kv_his[p] = base_pos + min(tile * 32 + 31, q_len - 1) + 1;",
        "const int first_pos = base_pos + tile * 32;",
        "const int piece = q * 4 + g;",
        "constexpr int PPW = L::PIECES / 4;",
        "const int gx = ((max_q_len + 31) / 32 + 1) / 2, gy = Hkv * ((G + 3) / 4);",
    ):
        assert line in src, f"kernel no longer contains: {line}"