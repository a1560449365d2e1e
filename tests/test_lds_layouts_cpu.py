"""The LDS layouts of the hot kernels against the bank model of MI355X_MICROARCH.md (section LDS), by enumeration -- no GPU.

DESIGN.md claims the matrix-core kernels' fragment reads and conversion stores, and the AES kernels' replicated tables, are free
of bank conflicts (and the PMC passes in profiles/ show SQ_LDS_BANK_CONFLICT = 0 for them).  The address formulas below restate the
kernels' (file and lambda named next to each); the model is the guide's: a wave64 access is served in fixed lane groups, one LDS
cycle per group when no two lanes of the group with DIFFERENT addresses touch the same bank (identical addresses broadcast).
    ds_read_b128   4 groups of 16 lanes {0-3,12-15,20-27} {4-11,16-19,28-31} {32-35,44-47,52-59} {36-43,48-51,60-63}; 64 banks
    ds_read_b64    2 groups of 32 lanes; 64 banks          ds_read_b32 / ds_write_b32   2 groups of 32 lanes; 32 banks
    ds_write_b32   a 2-way conflict costs nothing extra (the data transfer, not the array, sets its 4 cycles)"""
import itertools

import numpy as np
import pytest

G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
        list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
G32 = [list(range(0, 32)), list(range(32, 64))]


def worst_multiplicity(addr, nbytes, groups, nbanks):
    """max over lane groups and banks of the number of DISTINCT addresses that touch the bank (1 = conflict-free)"""
    worst = 0
    for grp in groups:
        per_bank = {}
        for lane in grp:
            a = int(addr[lane])
            assert a % min(nbytes, 16) == 0, "misaligned access"
            for w in range(nbytes // 4):
                per_bank.setdefault((a // 4 + w) % nbanks, set()).add(a)
        worst = max(worst, max(len(v) for v in per_bank.values()))
    return worst


N, KS = 256, 64
LANE = np.arange(64)
L16, KG = LANE & 15, LANE >> 4


def col_row(c):
    return ((c & 7) << 1) | (c >> 3)


def col_swz(c):
    return (c >> 1) & 3


def test_mm_fragment_reads_are_conflict_free():
    """mm_mfma_blk2_kernel.hip / mm_mfma_blk3_kernel.hip: aOff, panelA (f panel, 256-byte rows, 16-byte slots XORed with the row) and
    bOff (s slab: plane[q][row(c)][64 B], slots XORed with (c / 2) % 4), every k-slab position, every row block"""
    for slab in range(4):
        for rb in range(4):
            a = (L16 * N + ((KG ^ L16) * 16)) ^ (slab * 64)
            assert worst_multiplicity(a + rb * 16 * N, 16, G128, 64) == 1, ("A", slab, rb)
    b = col_row(L16) * KS + ((KG ^ col_swz(L16)) * 16)
    for q in range(4):
        assert worst_multiplicity(b + q * 16 * KS, 16, G128, 64) == 1, ("B", q)


def test_mm_conversion_stores_stay_within_two_way():
    """the s conversion (dstB: staging round u, column half h, plane q) and the background f panel pieces (panelDst) are
    ds_write_b32: up to 2-way is free"""
    col2 = 2 * (LANE & 7)
    dst0 = col_row(col2) * KS + ((((LANE >> 3) >> 2) ^ col_swz(col2)) * 16) + ((LANE >> 3) & 3) * 4
    for u, h, q in itertools.product(range(2), range(2), range(4)):
        a = (dst0 ^ (u * 32)) + h * 2 * KS + q * 16 * KS
        assert worst_multiplicity(a, 4, G32, 32) <= 2, ("s", u, h, q)
    # blk2: 512 threads, piece j of a panel: row 8 j + wave, k-quad = lane; four plane stores 64 KiB / 4 apart
    for wv, j, p in itertools.product(range(8), range(8), range(4)):
        d0 = wv * N + (((LANE >> 2) ^ wv) * 16) + (LANE & 3) * 4
        a = (d0 ^ ((j & 1) * 128)) + j * 8 * N + p * 64 * N
        assert worst_multiplicity(a, 4, G32, 32) <= 2, ("f", wv, j, p)


def test_mm_wide_staging_writes_the_same_image_without_conflicts():
    """mm_mfma_blk3_kernel.hip, round 5 (COAST_MM3_WIDE): a lane owns four adjacent columns x four k of a slab (lane -> column quad l % 4,
    k-quad l / 4; four buffer_load_dwordx4 per slab).  Its conversion stores must hit exactly the bytes the fragment reads expect -- column c,
    k..k+3 at colRow(c) * 64 + ((k / 16) ^ colSwz(c)) * 16 + k % 16 of a plane -- every (column, k-quad) once, and stay conflict-free."""
    cq, kquad = LANE & 3, LANE >> 2
    dst0 = (8 * (LANE & 1) + ((LANE & 3) >> 1)) * KS + (((LANE >> 4) ^ (2 * (LANE & 1))) * 16) + ((LANE >> 2) & 3) * 4
    seen = set()
    for h in range(4):
        a = (dst0 ^ ((h >> 1) * 16)) + h * 2 * KS
        c, k = 4 * cq + h, 4 * kquad
        want = col_row(c) * KS + (((k >> 4) ^ col_swz(c)) * 16) + (k & 15)
        assert (a == want).all(), h
        seen.update((int(x), int(y)) for x, y in zip(c, k))
        for q in range(4):
            assert worst_multiplicity(a + q * 16 * KS, 4, G32, 32) == 1, ("wide", h, q)
    assert len(seen) == 16 * 16  # 16 columns x 16 k-quads: the whole slab
    # the loads: row 4 (l / 4) + kk, columns 4 (l % 4) .. + 3 -> one instruction = 16 rows x 64 contiguous bytes
    voff = ((4 * kquad) * N + 4 * cq) * 4
    for kk in range(4):
        rows = (voff + kk * N * 4) // (N * 4)
        assert sorted(set(rows.tolist())) == [4 * r + kk for r in range(16)]
        for r in set(rows.tolist()):
            cols = sorted(((voff + kk * N * 4) % (N * 4))[rows == r].tolist())
            assert cols == [0, 16, 32, 48]


@pytest.mark.parametrize("nrep", [1, 2, 3])
def test_aes_replicated_tables_never_conflict(nrep):
    """aes_kernel.hip aes_rep_addr (round 3): entry value v owns a 256-byte row of a 64 KiB table block -- four slots of 16 copies x 4
    bytes (decryption block 1: 16 copies x 8-byte pairs in slots 0-1, rsbox dwords in slot 2) -- and the lanes of block slot q read
    copy q % 16 (lane % 16 unprotected).  Random table indices per block -- the replicas of a block look up the same entry.  DWC and
    TMR: conflict-free whatever the indices; unprotected: 32 blocks share 16 copies, two-way."""
    rng = np.random.default_rng(nrep)
    q = LANE // nrep if nrep > 1 else LANE
    copy = q & 15
    bound = 1 if nrep > 1 else 2
    for _ in range(2000):
        v_block = rng.integers(0, 256, 64)
        v = v_block[q]  # every replica lane of a block holds the same byte when nothing is upset
        for r in range(4):
            assert worst_multiplicity(v * 256 + r * 64 + copy * 4, 4, G32, 32) <= bound
        assert worst_multiplicity(65536 + v * 256 + copy * 8, 8, G32, 64) <= bound
        assert worst_multiplicity(65536 + v * 256 + copy * 8 + 4, 4, G32, 32) <= bound  # the S-box half of a pair alone
        assert worst_multiplicity(65536 + v * 256 + 128 + copy * 4, 4, G32, 32) <= bound  # rsbox
    for v0 in (0, 1, 254, 255):
        v = np.full(64, v0)
        assert worst_multiplicity(v * 256 + copy * 4, 4, G32, 32) <= bound


def test_aes_lookup_address_is_one_byte_permute():
    """the address of a lookup = the byte string {copy offset, x.byte[B], block, 0}: what v_perm_b32(x, laneSel, selector) returns for
    the selectors of aes_rep_addr (bytes 0..3 of the result picked from {laneSel bytes 0..3, x bytes 0..3} = indices 0..7, 0x0c = 0)"""
    def perm(a, b, sel):
        src = [(b >> (8 * i)) & 0xff for i in range(4)] + [(a >> (8 * i)) & 0xff for i in range(4)]
        out = 0
        for k in range(4):
            idx = (sel >> (8 * k)) & 0xff
            out |= (0 if idx == 0x0c else src[idx]) << (8 * k)
        return out

    rng = np.random.default_rng(5)
    for _ in range(2000):
        x, copy = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 16))
        for B in range(4):
            byte = (x >> (8 * B)) & 0xff
            for blk, lane_sel in ((0, copy * 4 | 0x100), (1, copy * 4 | 0x100), (1, copy * 8 | 0x100)):
                sel = (0x0c010000 if blk else 0x0c0c0000) + ((4 + B) << 8)
                assert perm(x, lane_sel, sel) == blk * 65536 + byte * 256 + (lane_sel & 0xff)


def test_aes_table_fill_is_conflict_free():
    """the fill loops of aes128_enc_rep_kernel / aes128_dec_rep_kernel: a wave writes whole rows, lane = (slot r, copy c): at most two
    lanes per bank (free for ds_write_b32); the pair stores of decryption block 1 come from 16 lanes"""
    c, r = LANE & 15, (LANE >> 4) & 3
    for v in (0, 7, 255):
        assert worst_multiplicity(v * 256 + r * 64 + c * 4, 4, G32, 32) <= 2
    pair = (65536 + 3 * 256 + (LANE & 15) * 8)[:16]
    assert len(set((pair // 4) % 64)) == 16


def test_aes_table_identities():
    """what lets the replicated kernels drop the byte tables and three of the four Tis tables: with Te_0[v] = (2S, S, S, 3S) and
    Td_0 / Tis_0 = (14, 9, 13, 11) x rsbox[v] / S[v] as little-endian bytes, table r is table 0 rotated left by r bytes, and
    S[v] is byte 1 of Te_0[v] (aes_kernel.hip aes_tables_kernel, aes_pick_b1, aes_rotl8)"""
    from oracle import oracle as orc

    def xt(a):
        return ((a << 1) ^ (0x1B if a & 0x80 else 0)) & 0xFF

    def mul(a, b):
        p = 0
        for _ in range(8):
            if b & 1:
                p ^= a
            a, b = xt(a), b >> 1
        return p

    # S-box from one-byte encryptions through the oracle's plain AES would need the key schedule; use the definition instead
    inv = [0] + [next(y for y in range(1, 256) if mul(x, y) == 1) for x in range(1, 256)]
    sbox = []
    for x in range(256):
        s = inv[x]
        for k in range(1, 5):
            s ^= ((inv[x] << k) | (inv[x] >> (8 - k))) & 0xFF
        sbox.append(s ^ 0x63)
    assert sbox[0] == 0x63 and sbox[0x53] == 0xED  # FIPS-197 figure 7
    ct, _last_round_key = orc.aes128_plain(bytes(16), bytes(16), 0)
    assert ct.hex() == "66e94bd4ef8a2c3b884cfa59ca342b2e"  # the oracle's cipher agrees with the published all-zero vector

    def pack(b):
        return b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24)

    def rotl(x, nbytes):
        n = 8 * nbytes
        return ((x << n) | (x >> (32 - n))) & 0xFFFFFFFF if n else x

    rsbox = [0] * 256
    for x, s in enumerate(sbox):
        rsbox[s] = x
    for v in range(256):
        s = sbox[v]
        s2, s3 = xt(s), xt(s) ^ s
        te = [pack([s2, s, s, s3]), pack([s3, s2, s, s]), pack([s, s3, s2, s]), pack([s, s, s3, s2])]
        assert all(te[r] == rotl(te[0], r) for r in range(4)) and (te[0] >> 8) & 0xFF == s
        for w in (rsbox[v], s):
            e, b, d, n = mul(w, 14), mul(w, 11), mul(w, 13), mul(w, 9)
            t = [pack([e, n, d, b]), pack([b, e, n, d]), pack([d, b, e, n]), pack([n, d, b, e])]
            assert all(t[r] == rotl(t[0], r) for r in range(4))


def _expected_cycles(addr_of_lane, trials, rng, nbanks=32):
    """mean LDS-array cycles of one wave-lookup: per 32-lane group, the largest number of distinct dwords on one bank"""
    tot = 0.0
    for _ in range(trials):
        addr = addr_of_lane(rng)
        for grp in G32:
            per_bank = {}
            for lane in grp:
                per_bank.setdefault((int(addr[lane]) // 4) % nbanks, set()).add(int(addr[lane]) // 4)
            tot += max(len(v) for v in per_bank.values())
    return tot / trials


def test_bank_model_reproduces_the_measured_conflict_cycles():
    """the model against the PMC passes in profiles/: SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS of the kernels whose table indices are
    random -- crc16_stream_kernel<3,2,true> 4.08 cycles per lookup (r02_crc16_256_rocprofv3_summary.txt: 8.54e8 / 2.094e8; 21
    blocks per wave, 2-byte entries of a 64 Ki-entry table), aes128_enc_fast_kernel<2> 4.14 (r02_aes_rocprofv3_summary.txt: 2.72e7 /
    6.57e6; 32 blocks per wave, 147 dword + 56 byte lookups per block) -- and the replicated AES tables' 2.19 (r02d_aes: 1.54e7 /
    7.04e6).  The random-index costs are what the layouts above remove for AES and what the crc16 stream has to live with."""
    rng = np.random.default_rng(0)
    q3 = np.minimum(LANE // 3, 20)  # the idle 64th lane repeats its neighbour's address
    crc = _expected_cycles(lambda r: r.integers(0, 65536, 21)[q3] * 2, 4000, rng)
    assert abs(crc - 4.08) / 4.08 < 0.06, crc
    q2 = LANE // 2
    dword = _expected_cycles(lambda r: r.integers(0, 256, 32)[q2] * 4, 4000, rng)
    byte = _expected_cycles(lambda r: r.integers(0, 256, 32)[q2], 4000, rng)
    classic = (147 * dword + 56 * byte) / 203
    assert abs(classic - 4.14) / 4.14 < 0.08, (dword, byte, classic)
    rep = _expected_cycles(lambda r: (r.integers(0, 256, 32)[q2] * 16 + (q2 & 15)) * 4, 2000, rng)
    assert rep == 2.0  # + the table fill and the counters' atomics in the measured 2.19
