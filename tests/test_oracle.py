"""CPU tests: the oracle against every golden vector the reference's tests hold for the hot path
(SURVEY.md section 8c), against the reference itself (oracle/_ref, when built), and the voter/counter rules."""
import hashlib

import numpy as np
import pytest

from conftest import gen_mm


@pytest.mark.parametrize("n", [9, 19, 30, 32])
def test_mm_plain_matches_reference_fixtures(orc, golden, n):
    f, s, r = golden["mm"]["f%d" % n], golden["mm"]["s%d" % n], golden["mm"]["r%d" % n]
    got = orc.mm_plain(f, s)
    assert (got == r).all()
    assert orc.mm_xor(got) == golden["mm_xor_golden_%d" % n]  # mm_tmr.c:31, mm.inc:43, mm.inc:65


def test_mm_generator_equivalence_and_256(orc, golden):
    for n in (9, 19, 30):
        f, s = gen_mm(n)
        assert (f == golden["mm"]["f%d" % n]).all() and (s == golden["mm"]["s%d" % n]).all()
    f, s = gen_mm(256)
    assert hashlib.sha256(f.tobytes() + s.tobytes()).hexdigest() == golden["mm_inputs_sha256_256"]
    r = orc.mm_plain(f, s)
    assert orc.mm_xor(r) == golden["mm_xor_golden_256"] == 458951617
    assert hashlib.sha256(r.tobytes()).hexdigest() == golden["mm_result_sha256_256"]


def test_mm_lanl_variant(orc, golden):
    ij = np.fromfunction(lambda i, j: i * j, (32, 32), dtype=np.int64).astype(np.uint32)
    assert (orc.mm_plain(ij, ij) == golden["mm"]["lanl_r32"]).all()


@pytest.mark.parametrize("tag", ["10", "4000"])
def test_sha256_fixtures(orc, golden, tag):
    data = golden["sha"]["data" + tag].tobytes()
    assert orc.sha256_plain(data) == golden["sha"]["golden" + tag].tobytes()


def test_sha256_lengths_vs_hashlib(orc):
    rng = np.random.default_rng(1)
    for ln in list(range(0, 130)) + [255, 256, 1000]:
        d = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        assert orc.sha256_plain(d) == hashlib.sha256(d).digest(), ln


def test_aes_kat(orc, golden):
    # tests/aes/aes.c:91-99: enc(input,key)==ciphertext, dec(.,key2)==plaintext
    errors = 0
    for row in golden["aes_kat"]:
        key, key2, ct, pt, inp = (row[16 * q:16 * q + 16].tobytes() for q in range(5))
        enc, k_after = orc.aes128_plain(inp, key, 0)
        dec, k2_after = orc.aes128_plain(enc, key2, 1)
        errors += (inp != pt) + (enc != ct) + (dec != pt)
        assert k2_after == key2  # decrypt walks the key schedule back to the cipher key
    assert errors == 0
    assert hashlib.sha256(bytes(np.ctypeslib.as_array(orc.lib().orc_aes_sbox(), (256,)))).hexdigest() == golden[
        "aes_sbox_sha256"]


def test_crc16_vectors(orc, golden):
    for v in golden["crc16_vectors"]:
        assert orc.crc16_plain(bytes.fromhex(v["data"])) == v["crc"]
    assert orc.crc16_plain(b"Automated TMR") == 0x5BA3  # tests/crc16/crc16.c:14,40 compiled unmodified


def test_against_reference_build(orc, golden):
    if orc.ref() is None:
        pytest.skip("oracle/_ref not built (no reference checkout on this box)")
    rng = np.random.default_rng(7)
    for ln in (0, 1, 55, 56, 63, 64, 65, 119, 120, 300):
        d = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        assert orc.sha256_plain(d) == orc.ref_sha256(d)
    for _ in range(64):
        st, key = rng.integers(0, 256, 16, dtype=np.uint8).tobytes(), rng.integers(0, 256, 16, dtype=np.uint8).tobytes()
        for d in (0, 1):
            assert orc.aes128_plain(st, key, d) == orc.ref_aes(st, key, d)
    for ln in range(0, 256, 5):
        d = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        assert orc.crc16_plain(d) == orc.ref_crc16(d)
    f, s = gen_mm(32, seed=3)
    assert (orc.mm_plain(f, s) == orc.ref_mm(f, s, 0)[0]).all()


# ------------------------------------------------------------------ voter / counter semantics
def test_tmr_clean_run_counts(orc, golden):
    f, s = golden["mm"]["f9"], golden["mm"]["s9"]
    r, st, _ = orc.mm_xmr(f, s, replicas=3)
    assert (r[0] == golden["mm"]["r9"]).all()
    assert st == {"errors_corrected": 0, "sync_count": 81, "dwc_detected": 0}
    r, st, _ = orc.mm_xmr(f, s, replicas=3, sync_every=4)  # + votes after k=4,8 (k+1<n)
    assert st["sync_count"] == 81 * 3 and st["errors_corrected"] == 0
    r, st, _ = orc.mm_xmr(f, s, replicas=1)
    assert st["sync_count"] == 0 and (r[0] == golden["mm"]["r9"]).all()


def test_tmr_voter_is_select_not_bitwise_majority(orc, golden):
    """synchronization.cpp:934-938: vote = (a==b) ? a : c on the whole value."""
    f, s, good = golden["mm"]["f9"], golden["mm"]["s9"], golden["mm"]["r9"]
    item = 9 * 4 + 5
    # single fault in any one replica: corrected, +1
    for rep in range(3):
        r, st, _ = orc.mm_xmr(f, s, faults=orc.make_faults([(item, rep, orc.SITE_MM_ACC, 3, 7)]))
        assert (r[0] == good).all() and st["errors_corrected"] == 1
    # replicas 0 and 1 corrupted identically: the corrupted value wins, counted once (replica 2 differs)
    fl = orc.make_faults([(item, 0, orc.SITE_MM_ACC, 9, 7), (item, 1, orc.SITE_MM_ACC, 9, 7)])
    r, st, _ = orc.mm_xmr(f, s, faults=fl)
    assert r[0].reshape(-1)[item] == good.reshape(-1)[item] ^ (1 << 7) and st["errors_corrected"] == 1
    # replicas 0 and 2 corrupted identically (0 != 1): replica 2 is taken unconditionally -> corrupted output;
    # a bitwise-majority voter would ALSO give the corrupted value here, but for different bits it differs:
    fl = orc.make_faults([(item, 0, orc.SITE_MM_ACC, 9, 3), (item, 2, orc.SITE_MM_ACC, 9, 12)])
    r, st, _ = orc.mm_xmr(f, s, faults=fl)
    assert r[0].reshape(-1)[item] == good.reshape(-1)[item] ^ (1 << 12)  # bitwise majority would give `good`
    assert st["errors_corrected"] == 1
    # never more than +1 per voted value
    fl = orc.make_faults([(item, 0, orc.SITE_MM_ACC, 9, 3), (item, 1, orc.SITE_MM_ACC, 9, 4),
                          (item, 2, orc.SITE_MM_ACC, 9, 5)])
    _, st, _ = orc.mm_xmr(f, s, faults=fl)
    assert st["errors_corrected"] == 1


def test_dwc_detects_and_keeps_replica0(orc, golden):
    f, s, good = golden["mm"]["f9"], golden["mm"]["s9"], golden["mm"]["r9"]
    item = 17
    r, st, det = orc.mm_xmr(f, s, replicas=2, faults=orc.make_faults([(item, 1, orc.SITE_MM_OPA, 2, 31)]))
    assert (r[0] == good).all() and st["dwc_detected"] == 1 and det[item] == 1 and det.sum() == 1
    r, st, det = orc.mm_xmr(f, s, replicas=2, faults=orc.make_faults([(item, 0, orc.SITE_MM_ACC, 9, 0)]))
    assert r[0].reshape(-1)[item] == good.reshape(-1)[item] ^ 1 and st["dwc_detected"] == 1


def test_fault_sites_other_kernels(orc, golden):
    rng = np.random.default_rng(3)
    # sha256: any single replica-private flip is corrected; digest unchanged
    msgs = rng.integers(0, 256, (6, 64), dtype=np.uint8)
    clean, st0, _ = orc.sha256_xmr(msgs, 64)
    assert st0["sync_count"] == 6 * (8 * 2 + 8) and st0["errors_corrected"] == 0
    for m in range(6):
        assert clean[m].tobytes() == hashlib.sha256(msgs[m].tobytes()).digest()
    fl = orc.make_faults([(0, 0, orc.SITE_SHA_M, 5, 1), (1, 1, orc.SITE_SHA_M, 64 + 40, 31),
                          (2, 2, orc.SITE_SHA_WV, 63, 9, 4), (3, 1, orc.SITE_SHA_STATE, 1, 0, 7),
                          (4, 0, orc.SITE_SHA_STATE, 2, 13, 3)])
    d, st, _ = orc.sha256_xmr(msgs, 64, faults=fl)
    assert (d == clean).all() and st["errors_corrected"] >= 5
    # aes DWC: flip detected, clean otherwise
    stt, key = rng.integers(0, 256, (4, 16), dtype=np.uint8), rng.integers(0, 256, (4, 16), dtype=np.uint8)
    c, k, st, det = orc.aes128_xmr(stt, key, 0)
    assert st["dwc_detected"] == 0 and st["sync_count"] == 4 * 8
    c2, k2, st, det = orc.aes128_xmr(stt, key, 0, faults=orc.make_faults([(2, 1, orc.SITE_AES_STATE, 4, 17, 2)]))
    assert st["dwc_detected"] == 1 and list(det) == [0, 0, 1, 0] and (c2 == c).all()
    # aes TMR corrects
    c3, k3, st, _ = orc.aes128_xmr(stt, key, 0, replicas=3, faults=orc.make_faults([(2, 0, orc.SITE_AES_KEY, 4, 17, 2)]))
    assert (c3 == c).all() and (k3 == k).all() and st["errors_corrected"] >= 1
    # crc16: bits >= 16 of the crc register are dead (legal no-effect hit, injector.py:202-207 flips any of 32)
    data = rng.integers(0, 256, (5, 64), dtype=np.uint8)
    crc, st, _ = orc.crc16_xmr(data, 64)
    assert st["sync_count"] == 5 and [orc.crc16_plain(r.tobytes()) for r in data] == list(crc)
    crc2, st, _ = orc.crc16_xmr(data, 64, faults=orc.make_faults([(1, 2, orc.SITE_CRC_CRC, 10, 20),
                                                                   (3, 0, orc.SITE_CRC_X, 63, 2)]))
    assert (crc2 == crc).all() and st["errors_corrected"] == 1


def test_cpu_tmr_baseline_matches(orc, golden):
    for n in (9, 30):
        f, s = golden["mm"]["f%d" % n], golden["mm"]["s%d" % n]
        r, err, cnt, syncs = orc.cpu_tmr_mm(f, s, golden["mm_xor_golden_%d" % n])
        assert (r == golden["mm"]["r%d" % n]).all() and err == 0 and cnt == 0
        # (n+1)(n^2+n+1) loop-condition votes in matrix_multiply + checkGolden's n^2+1 + return
        assert syncs == (n + 1) * (n * n + n + 1) + n * n + 1 + 1


def test_default_mode_exit_vote_semantics(orc):
    """orc_sync_copies: word-wise (a==b)?a:c with +1 per differing word, scrub re-converges the copies; DWC compares."""
    a = np.arange(10, dtype=np.uint32)
    c = [a.copy(), a.copy(), a.copy()]
    c[1][3] ^= 1 << 7            # single copy hit: corrected
    c[0][5] ^= 4
    c[1][5] ^= 4                 # copies 0 and 1 hit identically: the corrupted value wins, counted once
    c[0][8] ^= 2
    c[2][8] ^= 16                # 0 != 1 -> copy 2 is taken unconditionally
    voted, after, st, det = orc.sync_copies(c)
    exp = a.copy()
    exp[5] ^= 4
    exp[8] ^= 16
    assert (voted == exp).all() and st == {"errors_corrected": 3, "sync_count": 10, "dwc_detected": 0}
    assert list(np.nonzero(det)[0]) == [3, 5, 8] and all((x == exp).all() for x in after)
    voted, after, st, det = orc.sync_copies(c[:2])
    assert st == {"errors_corrected": 0, "sync_count": 10, "dwc_detected": 2} and (voted == c[0]).all()


def test_no_store_data_sync_semantics(orc):
    """-noStoreDataSync (synchronization.cpp:197-224, :324): the final store of r[i][j] is not voted -- replica 0's word is
    stored, upsets in the other replicas vanish, nothing is counted; loop-condition votes (sync_every) are untouched;
    crc16 has no store sync point at all (its result is a return value)."""
    import coast_amd

    def t3(st):
        return (st["errors_corrected"], st["sync_count"], st["dwc_detected"])

    rng = np.random.default_rng(3)
    n = 9
    f = rng.integers(0, 2**32, (1, n, n), dtype=np.uint32)
    s = rng.integers(0, 2**32, (1, n, n), dtype=np.uint32)
    clean, st0, _ = orc.mm_xmr(f, s)
    assert t3(st0) == (0, n * n, 0)
    F = 1
    r, st, det = orc.mm_xmr(f, s, flags=F)
    assert (r == clean).all() and t3(st) == (0, 0, 0)
    hit1 = coast_amd.make_faults([(5, 1, 0, 3, 7)])      # replica 1: invisible
    r, st, det = orc.mm_xmr(f, s, faults=hit1, flags=F)
    assert (r == clean).all() and t3(st) == (0, 0, 0) and not det.any()
    hit0 = coast_amd.make_faults([(5, 0, 0, 3, 7)])      # replica 0: silent data corruption
    r, st, det = orc.mm_xmr(f, s, faults=hit0, flags=F)
    assert r.reshape(-1)[5] != clean.reshape(-1)[5] and (np.delete(r.reshape(-1), 5) == np.delete(clean.reshape(-1), 5)).all()
    assert t3(st) == (0, 0, 0)
    r, st, det = orc.mm_xmr(f, s, sync_every=2, faults=hit0, flags=F)   # the loop vote after k=3 repairs it
    assert (r == clean).all() and t3(st)[0] == 1 and t3(st)[1] == n * n * ((n - 1) // 2)
    data = rng.integers(0, 256, (4, 50), dtype=np.uint8)
    fl = coast_amd.make_faults([(2, 0, 24, 10, 3)])
    a = orc.crc16_xmr(data, 50, faults=fl, flags=F)
    b = orc.crc16_xmr(data, 50, faults=fl)
    assert (a[0] == b[0]).all() and t3(a[1]) == t3(b[1]) == (1, 4, 0)


def test_cache_test_golden_and_protection_model(orc):
    """calc_sum (tests/cache_test/cacheTest.c:101-177): the reference's own golden (generateGolden, :88: 179700 for 600
    elements), the scrub, and the sync-point arithmetic of the restated protection."""
    import coast_amd

    n = 600
    a = np.tile(np.arange(n, dtype=np.int32), (5, 1))
    out, sums, nerrs, st, det = orc.cache_test_xmr(a)
    assert (sums == 179700).all() and (nerrs == 0).all() and (out == a).all()
    assert st == {"errors_corrected": 0, "sync_count": 5 * (n + 2), "dwc_detected": 0}   # n conditions + sum + count
    b = a.copy()
    b[1, 7] = 99          # memory upsets: every replica sees them -> the branch is taken, the element is rewritten
    b[3, 0] = -5
    b[3, 599] = 0
    out, sums, nerrs, st, det = orc.cache_test_xmr(b)
    assert (out == a).all() and nerrs.tolist() == [0, 1, 0, 2, 0] and st["errors_corrected"] == 0 and not det.any()
    assert sums.tolist() == [179700, 179700 + 92, 179700, 179700 - 5 - 599, 179700]    # the sum sees the array as found
    for rep in (2, 1):
        o2, s2, e2, st2, _ = orc.cache_test_xmr(b, replicas=rep)
        assert (o2 == out).all() and (s2 == sums).all() and (e2 == nerrs).all()
        assert st2["sync_count"] == (5 * (n + 2) if rep == 2 else 0)
    # register upsets in one replica: out-voted, counted once per unequal vote
    fl = coast_amd.make_faults([(0, 1, 33, 5, 3),      # loaded array[5] of replica 1: its condition AND its sum go wrong -> 2 votes
                                (2, 0, 32, 100, 9),    # running sum of replica 0 -> the return-value vote
                                (4, 2, 34, 600, 0)])   # numberOfErrors of replica 2 after the loop -> the stored-count vote
    out, sums, nerrs, st, det = orc.cache_test_xmr(a, faults=fl)
    assert (sums == 179700).all() and (nerrs == 0).all() and (out == a).all()
    assert st["errors_corrected"] == 4 and det.tolist() == [1, 0, 1, 0, 1]
    # the same upsets unprotected: silent corruption (replica 0 is the only copy)
    fl0 = coast_amd.make_faults([(0, 0, 33, 5, 3), (2, 0, 32, 100, 9)])
    out, sums, nerrs, st, det = orc.cache_test_xmr(a, replicas=1, faults=fl0)
    assert sums[0] != 179700 and nerrs[0] == 1 and sums[2] != 179700
    # DWC: flagged, never corrected
    out, sums, nerrs, st, det = orc.cache_test_xmr(a, replicas=2, faults=coast_amd.make_faults([(3, 1, 32, 10, 4)]))
    assert st["dwc_detected"] == 1 and det.tolist() == [0, 0, 0, 1, 0]


def test_chstone_sha_golden_and_protection_model(orc, golden):
    """CHStone sha (tests/chstone/sha): the benchmark's own input and expected digest (sha_driver.c:49-50), outputs of the
    reference run in the build container on random inputs, and the restated protection (five store-data votes per
    sha_transform)."""
    import coast_amd

    ch = golden["chsha"]
    msg = ch["indata"].reshape(1, -1)                       # sha_stream: 2 x 8192 bytes, one running hash
    dig, st, det = orc.chsha_xmr(msg, 16384)
    assert dig[0].tolist() == ch["outData"].tolist() == golden["chsha_outData"]
    assert st == {"errors_corrected": 0, "sync_count": 5 * 257, "dwc_detected": 0}   # 256 data blocks + the padding block
    for q in range(6):
        d = ch["rand%d" % q]
        m = np.zeros((1, max(d.size, 64)), dtype=np.uint8)
        m[0, :d.size] = d
        assert orc.chsha_xmr(m, d.size, replicas=1)[0][0].tolist() == ch["rand%d_digest" % q].tolist()
    if orc.ref() is not None:                               # build container: against the reference itself, more inputs
        rng = np.random.default_rng(1)
        for ln in (64, 320, 2048):
            d = rng.integers(0, 256, ln, dtype=np.uint8)
            assert orc.chsha_xmr(d.reshape(1, -1), ln)[0][0].tolist() == orc.ref_chsha(d.tobytes()).tolist()
    # protection model on a 3-block message
    rng = np.random.default_rng(2)
    msgs = rng.integers(0, 256, (4, 192), dtype=np.uint8)
    clean, st0, _ = orc.chsha_xmr(msgs, 192)
    assert st0["sync_count"] == 4 * 5 * 4
    fl = coast_amd.make_faults([(0, 1, 40, 80 + 17, 5),        # W[17] of transform 1, replica 1
                                (1, 0, 41, 3 * 80 + 79, 31, 4),  # E before the last round of the padding transform, replica 0
                                (2, 2, 42, 2, 0, 3)])          # sha_info_digest[3] before transform 2, replica 2
    dig, st, det = orc.chsha_xmr(msgs, 192, faults=fl)
    assert (dig == clean).all() and det.tolist() == [1, 1, 1, 0] and st["errors_corrected"] >= 3
    dig1, _, _ = orc.chsha_xmr(msgs, 192, replicas=1, faults=coast_amd.make_faults([(1, 0, 41, 3 * 80 + 79, 31, 4)]))
    assert (dig1[1] != clean[1]).any() and (dig1[[0, 2, 3]] == clean[[0, 2, 3]]).all()
    _, st2, det2 = orc.chsha_xmr(msgs, 192, replicas=2, faults=coast_amd.make_faults([(3, 1, 40, 5, 9)]))
    assert st2["dwc_detected"] == 1 and det2.tolist() == [0, 0, 0, 1]


def test_quicksort_oracle_pinned_on_the_reference(orc):
    """tests/quicksort/quicksort.c has no golden in tree (it sorts rand() data and compares two of its own sorts): the oracle's
    restatement is pinned on the reference's quick_sort compiled from where it lies (oracle/_ref) and on sortedness, on the
    benchmark's size (580) and the edge shapes; the protected model leaves clean runs untouched in every mode / flag set."""
    rng = np.random.default_rng(580)
    cases = [rng.integers(-2**31, 2**31, 580, dtype=np.int64).astype(np.int32) for _ in range(8)]
    cases += [np.arange(580, dtype=np.int32), np.arange(580, dtype=np.int32)[::-1].copy(), np.zeros(580, np.int32),
              rng.integers(0, 3, 580).astype(np.int32), np.array([5], np.int32), np.array([2, 1], np.int32),
              np.array([1, 2, 1], np.int32), np.array([-2**31, 2**31 - 1, 0, -1], np.int32)]
    have_ref = orc.ref() is not None and hasattr(orc.ref(), "ref_quicksort")
    for a in cases:
        want = np.sort(a)
        assert (orc.quicksort_plain(a) == want).all()
        if have_ref:
            assert (orc.ref_quicksort(a) == want).all()
    batch = np.stack(cases[:12])
    for replicas in (3, 2, 1):
        for flags in (0, 1, 8, 16, 1 | 8 | 16):
            s, st, det, status = orc.quicksort_xmr(batch, replicas=replicas, flags=flags)
            assert (s == np.sort(batch, axis=1)).all() and not status.any() and not det.any()
            assert st["errors_corrected"] == 0 and (st["sync_count"] > 0) == (replicas > 1)
    # every class of sync point is counted: dropping a class lowers __SYNC_COUNT
    full = orc.quicksort_xmr(batch)[1]["sync_count"]
    assert full > orc.quicksort_xmr(batch, flags=8)[1]["sync_count"] > orc.quicksort_xmr(batch, flags=8 | 16)[1]["sync_count"] \
        > orc.quicksort_xmr(batch, flags=1 | 8 | 16)[1]["sync_count"] > 0


def test_quicksort_oracle_fault_semantics(orc):
    """a single upset in one replica never reaches the array under TMR; unprotected, the same upsets break the order or hang the
    sort (watchdog) -- the reason the benchmark exists (quicksort.c:52-58)"""
    rng = np.random.default_rng(1)
    a = rng.integers(-2**31, 2**31, (150, 580), dtype=np.int64).astype(np.int32)
    rows = [(k, int(rng.integers(0, 3)), int(rng.integers(48, 53)), int(rng.integers(0, 12000)), int(rng.integers(0, 32)))
            for k in range(150)]
    fl = np.zeros(len(rows), dtype=orc.FAULT_DTYPE)
    for q, (item, rep, site, step, bit) in enumerate(rows):
        fl[q] = (item, step, rep, site, bit, 0)
    s, st, det, status = orc.quicksort_xmr(a, faults=fl)
    assert (s == np.sort(a, axis=1)).all() and not status.any() and det.sum() > 50 and st["errors_corrected"] >= det.sum()
    fl0 = fl.copy()
    fl0["replica"] = 0
    s, st, det, status = orc.quicksort_xmr(a, replicas=1, faults=fl0)
    assert not (s == np.sort(a, axis=1)).all(axis=1).all() and (status == 1).any()
    s2, st2, det2, _ = orc.quicksort_xmr(a, replicas=2, faults=fl[fl["replica"] < 2])
    assert st2["dwc_detected"] == det2.sum() > 30


def test_quicksort_oracle_vs_committed_reference_vectors(orc):
    import os

    fx = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quicksort_fixtures.npz")))
    for q in range(6):
        assert (orc.quicksort_plain(fx["in%d" % q]) == fx["out%d" % q]).all()
        s, _, _, status = orc.quicksort_xmr(fx["in%d" % q][None])
        assert (s[0] == fx["out%d" % q]).all() and not status.any()


# ---- CHStone aes (tests/chstone/aes): Rijndael, nine key / block sizes ----
def _chaes_fx():
    import os

    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chaes_fixtures.npz")))


def test_chaes_oracle_matches_reference_vectors(orc):
    """outputs of the reference's own encrypt / decrypt (oracle/_ref, tests/golden/gen_golden.py) for all nine `type`s"""
    fx = _chaes_fx()
    for t in orc.CHAES_TYPES:
        out, st, det = orc.chaes_xmr(fx["st%d" % t], fx["key%d" % t], t, 0, replicas=3)
        nk, nb, nr = orc.chaes_geom(t)
        assert (out == fx["enc%d" % t]).all() and not det.any()
        assert st == {"errors_corrected": 0, "sync_count": 12 * nb, "dwc_detected": 0}
        back, _, _ = orc.chaes_xmr(out, fx["key%d" % t], t, 1, replicas=2)
        assert (back == fx["st%d" % t]).all()
        dec, st, _ = orc.chaes_xmr(fx["st%d" % t], fx["key%d" % t], t, 1, replicas=1, sync_every=1)
        assert (dec == fx["dec%d" % t]).all() and st["sync_count"] == 0
        _, st, _ = orc.chaes_xmr(fx["st%d" % t], fx["key%d" % t], t, 0, replicas=3, sync_every=1)
        assert st["sync_count"] == 12 * nb * nr
    # the benchmark's own test vector: FIPS-197 Appendix B (aes_enc.c:77-80)
    assert fx["enc128128"][0].tobytes().hex() == "3925841d02dc09fbdc118597196a0b32"
    with pytest.raises(ValueError):
        orc.chaes_plain(np.zeros(16, np.uint8), np.zeros(16, np.uint8), 128129)


def test_chaes_oracle_votes_out_single_upsets(orc):
    from coast_amd import make_faults

    fx = _chaes_fx()
    rng = np.random.default_rng(3)
    for t in (128128, 192256, 256192):
        nk, nb, nr = orc.chaes_geom(t)
        rows = []
        for q in range(12):
            if q % 2:
                rows.append((q, int(rng.integers(0, 3)), orc.SITE_CHAES_STATE, int(rng.integers(0, nr + 2)), int(rng.integers(0, 32)),
                             int(rng.integers(0, nb))))
            else:
                rows.append((q, int(rng.integers(0, 3)), orc.SITE_CHAES_WORD, int(rng.integers(0, nb * (nr + 1))),
                             int(rng.integers(0, 32))))
        out, st, det = orc.chaes_xmr(fx["st%d" % t], fx["key%d" % t], t, 0, replicas=3, faults=make_faults(rows))
        assert (out == fx["enc%d" % t]).all() and det.sum() >= 10 and st["errors_corrected"] >= det.sum()
        out1, _, _ = orc.chaes_xmr(fx["st%d" % t], fx["key%d" % t], t, 0, replicas=1, faults=make_faults([(r[0], 0) + r[2:] for r in rows]))
        assert (out1 != fx["enc%d" % t]).any(axis=1).sum() >= 10
        _, st2, det2 = orc.chaes_xmr(fx["st%d" % t], fx["key%d" % t], t, 0, replicas=2,
                                     faults=make_faults([(r[0], r[1] % 2) + r[2:] for r in rows]))
        assert st2["dwc_detected"] == det2.sum() >= 10


def test_cache_test_loop_counter_in_the_sor_schedule(orc):
    """ORC_F_BRANCH_SYNC / ORC_F_ADDR_SYNC for calc_sum (cacheTest.c:107-131): i beside sum / numberOfErrors inside the sphere of
    replication.  Clean arrays of n elements: n + 1 loop conditions, 2 n load offsets, n element compares, the two `if`s behind the loop,
    the returned sum and the stored error count = 4 n + 5 votes per array; a corrupt element adds its scrub's store offset and data and
    the report block's branches.  Results equal the
    default schedule's; a single upset of i is out-voted under TMR, detected under DWC."""
    n, na = 600, 7
    a = np.tile(np.arange(n, dtype=np.int32), (na, 1))
    B, A = 2, 4
    w_a, w_s, w_e, st, det = orc.cache_test_xmr(a, replicas=3, flags=B | A)
    assert st["sync_count"] == na * (4 * n + 5) and st["errors_corrected"] == 0 and not det.any()
    assert (w_s == 179700).all() and not w_e.any()                      # generateGolden(), cacheTest.c:86
    a[3, 17] = 99
    w_a, w_s, w_e, st, det = orc.cache_test_xmr(a, replicas=3, flags=B | A)
    # one corrupt element: its scrub's offset + data, the report block's `!first_error` / `!in_block`, the printf's array[i] offset,
    # and `local_errors == 0` behind the now wrong sum
    assert st["sync_count"] == na * (4 * n + 5) + 2 + 4 and w_e[3] == 1 and w_a[3, 17] == 17
    ref = orc.cache_test_xmr(a, replicas=3)
    assert (w_a == ref[0]).all() and (w_s == ref[1]).all() and (w_e == ref[2]).all()
    fl = orc.make_faults([(5, 1, 35, 100, 3)])                          # replica 1's i before condition 100
    t = orc.cache_test_xmr(a, replicas=3, flags=B | A, faults=fl)
    assert (t[0] == ref[0]).all() and (t[1] == ref[1]).all() and t[3]["errors_corrected"] > 0 and t[4][5] == 1
    dw = orc.cache_test_xmr(a, replicas=2, flags=B | A, faults=fl)
    assert dw[3]["dwc_detected"] == 1 and dw[4][5] == 1


def test_aes_loop_counters_in_the_sor_schedule(orc, golden):
    """ORC_F_BRANCH_SYNC / ORC_F_ADDR_SYNC for aes_enc_dec (TI_aes_128.c:107-235): `round` and `i` inside the sphere of replication.
    The counts follow from the source as written.  Encryption: branch conditions 1 (`if (dir)`) + 11 (`round < 10`) + per round
    1 (`if (dir)`) + 17 (`i < 16`) + 1 (`if (dir)`) + 13 (`i < 16` from 4) = 320, the MixColumns condition 3 + 8 x 4 + 3 = 38 operands,
    9 x (5 + 4) in its loop, 1 + 17 after the loop = 469; loop conditions alone 11 + 10 x 30 + 9 x 5 + 17 = 373.  Decryption: 593 / 514.
    The same numbers come out of the reference's own -O0 IR (tools/ir_sync_counts.py, tests/test_ir_counts_cpu.py).
    Results are the frozen schedule's -- the NIST vectors included."""
    B, A = 2, 4
    rng = np.random.default_rng(5)
    st = rng.integers(0, 256, (9, 16), dtype=np.uint8)
    ky = rng.integers(0, 256, (9, 16), dtype=np.uint8)
    for d, nbr, ngep in ((0, 469, 1818), (1, 593, 2516)):
        ref = orc.aes128_xmr(st, ky, d, replicas=3)
        b = orc.aes128_xmr(st, ky, d, replicas=3, flags=B)
        ba = orc.aes128_xmr(st, ky, d, replicas=3, flags=B | A)
        assert (b[0] == ref[0]).all() and (ba[0] == ref[0]).all() and (ba[1] == ref[1]).all()
        assert b[2]["sync_count"] == 9 * (8 + nbr) and ba[2]["sync_count"] == 9 * (8 + nbr + ngep)
        assert ba[2]["errors_corrected"] == 0
        # a single upset of a counter: out-voted under TMR with everything voted, flagged under DWC
        fl = orc.make_faults([(4, 1, 19, 100, 2), (7, 2, 18, 200, 0)])
        t = orc.aes128_xmr(st, ky, d, replicas=3, flags=B | A, faults=fl)
        assert (t[0] == ref[0]).all() and (t[1] == ref[1]).all() and t[2]["errors_corrected"] > 0 and t[3][4] == 1 and t[3][7] == 1
        dw = orc.aes128_xmr(st, ky, d, replicas=2, flags=B | A, faults=orc.make_faults([(4, 1, 19, 100, 2)]))
        assert dw[2]["dwc_detected"] == 1 and dw[3][4] == 1
    # the known-answer vectors through the counters-in-the-SoR walk
    kat = np.asarray(golden["aes_kat"], dtype=np.uint8)  # rows: key, key2, ciphertext, plaintext, input (tests/aes/aes.c:91-99)
    key, key2, ct, pt = (np.ascontiguousarray(kat[:, 16 * q:16 * q + 16]) for q in range(4))
    enc = orc.aes128_xmr(pt, key, 0, replicas=3, flags=B | A)
    assert (enc[0] == ct).all()
    dec = orc.aes128_xmr(ct, key2, 1, replicas=2, flags=B | A)
    assert (dec[0] == pt).all() and (dec[1] == key2).all()


def test_chsha_loop_counters_in_the_sor_schedule(orc, golden):
    """ORC_F_BRANCH_SYNC / ORC_F_ADDR_SYNC for CHStone sha (sha.c:84-172): sha_transform's i and sha_update's count inside the sphere
    of replication.  Per transform 17 + 65 + 4 x 21 = 166 loop conditions and 16 x 2 + 64 x 5 + 80 = 432 variable-index GEPs; per call
    the `while (count >= 64)` conditions (one per block + 1), the carry test and sha_final's `count > 56`.  Digests are the default
    schedule's -- the benchmark's own golden digest (2 x 8192 bytes, sha.h:59-60) included."""
    B, A = 2, 4
    fx = golden["chsha"]
    data = np.ascontiguousarray(fx["indata"].reshape(1, -1))
    ln = data.shape[1]
    nt = ln // 64 + 1
    ref = orc.chsha_xmr(data, ln, replicas=3)
    got = orc.chsha_xmr(data, ln, replicas=3, flags=B | A)
    assert (got[0] == ref[0]).all() and got[0][0].tolist() == fx["outData"].tolist()
    assert got[1]["sync_count"] == 5 * nt + (nt * 166 + nt + 2) + nt * 432 + 1 and got[1]["errors_corrected"] == 0  # (+1: sha_info_data[count++])
    fl = orc.make_faults([(0, 2, 43, 5000, 9), (0, 1, 44, 100, 30)])
    t = orc.chsha_xmr(data, ln, replicas=3, flags=B | A, faults=fl)
    assert (t[0] == ref[0]).all() and t[1]["errors_corrected"] > 0 and t[2][0] == 1
    dw = orc.chsha_xmr(data, ln, replicas=2, flags=B | A, faults=orc.make_faults([(0, 1, 43, 5000, 9)]))
    assert dw[1]["dwc_detected"] == 1


def test_counter_votes_are_what_stops_a_replica0_upset_aes_and_chsha(orc):
    """Unvoted uses follow the ORIGINAL instruction's operand = replica 0's copy (cloning.cpp:2247-2255): with the offset votes on, an
    upset of replica 0's loop counter is out-voted at the next GEP and the results stay right; under -noLoadSync -noStoreAddrSync the
    branch vote still corrects the direction, but the accesses of that iteration go where replica 0 points -- silent data corruption.
    (The GPU flag sweeps of tests/test_gpu_parity.py hold the kernels to the same outcomes.)"""
    B, A, NL, NS = 2, 4, 8, 16
    rng = np.random.default_rng(3)
    st = rng.integers(0, 256, (4, 16), dtype=np.uint8)
    ky = rng.integers(0, 256, (4, 16), dtype=np.uint8)
    ref = orc.aes128_xmr(st, ky, 0, replicas=3)
    for tick in (15, 30, 40):  # inside round 0's `state[i] = sbox[state[i] ^ key[i]]` loop
        fl = orc.make_faults([(1, 0, 19, tick, 1)])
        voted = orc.aes128_xmr(st, ky, 0, replicas=3, flags=B | A, faults=fl)
        loose = orc.aes128_xmr(st, ky, 0, replicas=3, flags=B | A | NL | NS, faults=fl)
        assert (voted[0] == ref[0]).all() and (voted[1] == ref[1]).all() and voted[2]["errors_corrected"] > 1
        assert not (loose[0][1] == ref[0][1]).all() and (loose[0][[0, 2, 3]] == ref[0][[0, 2, 3]]).all()
        assert loose[2]["errors_corrected"] == 1  # only the loop condition saw the disagreement
    msgs = rng.integers(0, 256, (3, 128), dtype=np.uint8)
    r2 = orc.chsha_xmr(msgs, 128, replicas=3)
    for tick in (5, 30, 120):
        fl = orc.make_faults([(1, 0, 43, tick, 2)])
        voted = orc.chsha_xmr(msgs, 128, replicas=3, flags=B | A, faults=fl)
        loose = orc.chsha_xmr(msgs, 128, replicas=3, flags=B | A | NL | NS, faults=fl)
        assert (voted[0] == r2[0]).all() and voted[1]["errors_corrected"] > 1
        assert not (loose[0][1] == r2[0][1]).all() and (loose[0][[0, 2]] == r2[0][[0, 2]]).all() and loose[1]["errors_corrected"] == 1


@pytest.mark.parametrize("seed", range(6))
def test_counters_in_the_sor_properties_across_kernels(orc, seed):
    """Properties every counters-in-the-SoR walk must have, on random inputs: (a) clean results equal the default schedule's for every
    flag set; (b) dropping a vote class never adds sync points and the classes add up (all = branch-only + address-only - the default
    schedule's votes); (c) one upset of one counter under TMR with everything voted leaves the results untouched and is counted; (d) the
    same upset under DWC flags the item."""
    rng = np.random.default_rng(9000 + seed)
    B, A, NL, NS = 2, 4, 8, 16
    st = rng.integers(0, 256, (6, 16), dtype=np.uint8)
    ky = rng.integers(0, 256, (6, 16), dtype=np.uint8)
    msgs = rng.integers(0, 256, (6, 128), dtype=np.uint8)
    arr = np.tile(np.arange(40, dtype=np.int32), (6, 1))
    arr[rng.integers(0, 6), rng.integers(0, 40)] = -7
    f = rng.integers(0, 2**32, (6, 5, 5), dtype=np.uint32)
    s2 = rng.integers(0, 2**32, (6, 5, 5), dtype=np.uint32)
    d = int(seed & 1)
    cases = {
        "aes": (lambda **k: orc.aes128_xmr(st, ky, d, **k), lambda r: np.concatenate([r[0], r[1]], axis=1), lambda r: r[2], lambda r: r[3],
                lambda: (int(rng.integers(0, 6)), int(rng.integers(1, 3)), int(rng.choice([18, 19])), int(rng.integers(0, 373)), int(rng.integers(0, 8)))),
        "chsha": (lambda **k: orc.chsha_xmr(msgs, 128, **k), lambda r: r[0], lambda r: r[1], lambda r: r[2],
                  lambda: (int(rng.integers(0, 6)), int(rng.integers(1, 3)), int(rng.choice([43, 44])), int(rng.integers(0, 500)), int(rng.integers(0, 32)))),
        "cache_test": (lambda **k: orc.cache_test_xmr(arr, **k), lambda r: np.concatenate([r[0], r[1][:, None], r[2][:, None].astype(np.int32)], axis=1),
                       lambda r: r[3], lambda r: r[4],
                       lambda: (int(rng.integers(0, 6)), int(rng.integers(1, 3)), 35, int(rng.integers(0, 41)), int(rng.integers(0, 32)))),
        "mm": (lambda **k: orc.mm_xmr(f, s2, **k), lambda r: r[0].reshape(6, -1), lambda r: r[1], lambda r: r[2].reshape(6, -1).any(axis=1),
               lambda: (25 * int(rng.integers(0, 6)), int(rng.integers(1, 3)), int(rng.choice([3, 4, 5])), int(rng.integers(0, 186)), int(rng.integers(0, 32)))),
    }
    for name, (run, outs, stats, det, fault) in cases.items():
        ref = run(replicas=3)
        base = stats(ref)["sync_count"]
        full = run(replicas=3, flags=B | A)
        nb, na = stats(run(replicas=3, flags=B))["sync_count"], stats(run(replicas=3, flags=A))["sync_count"]
        assert (outs(full) == outs(ref)).all(), name
        # (cache_test: the scrub stores the counter itself, a data vote that exists as soon as i is replicated -- in both single-class runs)
        extra = int((arr != np.arange(40, dtype=np.int32)).sum()) if name == "cache_test" else 0
        assert stats(full)["sync_count"] == nb + na - base - extra, name
        nl, ns = stats(run(replicas=3, flags=B | A | NL))["sync_count"], stats(run(replicas=3, flags=B | A | NS))["sync_count"]
        assert nl <= stats(full)["sync_count"] and ns <= stats(full)["sync_count"], name
        assert stats(run(replicas=3, flags=B | A | NL | NS))["sync_count"] == nb, name
        row = fault()
        fl = orc.make_faults([row])
        t = run(replicas=3, flags=B | A, faults=fl)
        assert (outs(t) == outs(ref)).all(), (name, row)
        if stats(t)["errors_corrected"]:
            item = row[0] // 25 if name == "mm" else row[0]
            assert bool(np.asarray(det(t))[item]), (name, row)
            dw = run(replicas=2, flags=B | A, faults=orc.make_faults([(row[0], 1) + row[2:]]))
            assert stats(dw)["dwc_detected"] == 1, (name, row)
