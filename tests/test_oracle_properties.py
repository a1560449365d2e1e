"""Property tests of the oracle's protection model (CPU only; hypothesis).  Whatever single-event upset the injector
describes -- any site, step, bit, replica -- the restated TMR must return the fault-free outputs, the restated DWC must
never corrupt silently, and an unprotected run must ignore faults addressed to replicas it does not have.  These are the
invariants the reference's fault-injection campaigns measure (jsonParser.py:162-186); no reference test asserts them, so
they are pinned here on the oracle that the GPU parity tests compare against."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

SET = settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])


def _mk(rows):
    import coast_amd

    return coast_amd.make_faults(rows)


class K:
    """one kernel of the oracle: run(replicas, faults) -> (outputs tuple of arrays, stats, detected)"""

    def __init__(self, name, run, sites, max_step, max_index, nitems):
        self.name, self.run, self.sites, self.max_step, self.max_index, self.nitems = name, run, sites, max_step, max_index, nitems


def _kernels(orc):
    rng = np.random.default_rng(12)
    f = rng.integers(0, 2**32, (2, 5, 5), dtype=np.uint32)
    s = rng.integers(0, 2**32, (2, 5, 5), dtype=np.uint32)
    msgs = rng.integers(0, 256, (3, 70), dtype=np.uint8)
    stt = rng.integers(0, 256, (3, 16), dtype=np.uint8)
    key = rng.integers(0, 256, (3, 16), dtype=np.uint8)
    data = rng.integers(0, 256, (3, 40), dtype=np.uint8)
    arr = np.tile(np.arange(40, dtype=np.int32), (3, 1))
    arr[1, 7] = -3  # one genuine memory upset for the scrub
    ch = rng.integers(0, 256, (3, 128), dtype=np.uint8)
    cst = rng.integers(0, 256, (3, 32), dtype=np.uint8)   # CHStone aes, 192-bit key, 256-bit block
    cky = rng.integers(0, 256, (3, 24), dtype=np.uint8)
    cprm = np.array([[42, 20, 10], [7, 13, 3], [99, 0, 5]], dtype=np.int32)

    def w(fn):
        def run(rep, fl):
            out = fn(rep, fl)
            return tuple(np.asarray(o) for o in out[:-2]), out[-2], out[-1]
        return run

    return [
        K("mm", w(lambda rep, fl: orc.mm_xmr(f, s, replicas=rep, sync_every=2, faults=fl)), [0, 1, 2], 5, 1, 50),
        K("sha256", w(lambda rep, fl: orc.sha256_xmr(msgs, 70, replicas=rep, faults=fl)), [8, 9, 10], 2 * 64, 8, 3),
        K("aes", w(lambda rep, fl: orc.aes128_xmr(stt, key, 0, replicas=rep, sync_every=1, faults=fl)), [16, 17], 10, 4, 3),
        K("crc16", w(lambda rep, fl: orc.crc16_xmr(data, 40, replicas=rep, sync_every=7, faults=fl)), [24, 25], 40, 1, 3),
        K("cache_test", w(lambda rep, fl: orc.cache_test_xmr(arr, replicas=rep, faults=fl)), [32, 33, 34], 40, 1, 3),
        K("chsha", w(lambda rep, fl: orc.chsha_xmr(ch, 128, replicas=rep, faults=fl)), [40, 41, 42], 3 * 80, 5, 3),
        # the statement-by-statement walks with every sync class on (COAST_F_BRANCH_SYNC | ADDR_SYNC | LOCAL_STORE_SYNC): the counters are
        # inside the sphere of replication and are themselves injection sites
        K("mm_walk", w(lambda rep, fl: orc.mm_xmr(f, s, replicas=rep, faults=[] if fl is None else _items(fl, 25), flags=WALK)),
          [0, 3, 4, 5], 6 * 31, 1, 2),
        K("sha256_walk", w(lambda rep, fl: orc.sha256_xmr(msgs, 70, replicas=rep, faults=fl, flags=WALK | 128)), [8, 9, 10, 11, 12], 72, 8, 3),
        K("aes_walk", w(lambda rep, fl: orc.aes128_xmr(stt, key, 1, replicas=rep, faults=fl, flags=WALK)), [16, 17, 18, 19], 560, 4, 3),
        K("crc16_walk", w(lambda rep, fl: orc.crc16_xmr(data, 40, replicas=rep, faults=fl, flags=WALK)), [24, 25, 26], 40, 1, 3),
        K("cache_test_walk", w(lambda rep, fl: orc.cache_test_xmr(arr, replicas=rep, faults=fl, flags=WALK)), [32, 33, 34, 35], 40, 1, 3),
        K("chsha_walk", w(lambda rep, fl: orc.chsha_xmr(ch, 128, replicas=rep, faults=fl, flags=WALK)), [40, 41, 42, 43, 44], 3 * 167, 5, 3),
        K("chaes_walk", w(lambda rep, fl: orc.chaes_xmr(cst, cky, 192256, 0, replicas=rep, faults=fl, flags=WALK)), [64, 65, 66, 67, 68], 900, 8, 3),
        K("crazycf_xmr", w(lambda rep, fl: orc.crazycf_xmr(cprm, rep, WALK, fl)), [72, 73, 74, 75], 70, 1, 3),
    ]


WALK = 2 | 4 | 64
NK = 14


def _items(fl, stride):
    """mm's walk addresses a CALL (item = b * n * n): move the generic rows' items onto the matrices"""
    fl = fl.copy()
    fl["item"] = (fl["item"] % 2) * stride
    return fl


fault = st.tuples(st.integers(0, 10**6), st.integers(0, 2), st.integers(0, 10**6), st.integers(0, 10**6), st.integers(0, 31),
                  st.integers(0, 7))


def _row(k, t, replica=None):
    item, rep, site, step, bit, index = t
    return (item % k.nitems, rep if replica is None else replica, k.sites[site % len(k.sites)], step % (k.max_step + 1), bit,
            index % k.max_index)


@pytest.mark.parametrize("ki", range(NK))
@SET
@given(t=fault)
def test_tmr_masks_any_single_upset(orc, ki, t):
    k = _kernels(orc)[ki]
    clean, st0, _ = k.run(3, None)
    out, st1, det = k.run(3, _mk([_row(k, t)]))
    assert all((a == b).all() for a, b in zip(out, clean)), k.name
    assert st1["sync_count"] == st0["sync_count"] and st1["dwc_detected"] == 0
    assert (st1["errors_corrected"] > 0) == bool(det.any())          # a correction is always attributed to its item
    assert det.sum() <= 1


@pytest.mark.parametrize("ki", range(NK))
@SET
@given(t=fault, u=fault)
def test_tmr_masks_two_upsets_in_the_same_replica(orc, ki, t, u):
    k = _kernels(orc)[ki]
    clean, _, _ = k.run(3, None)
    r = t[1]
    out, _, _ = k.run(3, _mk([_row(k, t, r), _row(k, u, r)]))
    assert all((a == b).all() for a, b in zip(out, clean)), k.name


@pytest.mark.parametrize("ki", range(NK))
@SET
@given(t=fault)
def test_dwc_never_corrupts_silently(orc, ki, t):
    k = _kernels(orc)[ki]
    clean, _, _ = k.run(2, None)
    row = _row(k, t, t[1] % 2)
    out, st1, det = k.run(2, _mk([row]))
    item = row[0] if k.name != "mm" else None
    differs = any((a != b).any() for a, b in zip(out, clean))
    if differs:
        assert st1["dwc_detected"] >= 1 and det.any(), k.name       # wrong output => the compare tripped
    assert st1["errors_corrected"] == 0                              # DWC detects, it never corrects


@pytest.mark.parametrize("ki", range(NK))
@SET
@given(t=fault)
def test_unprotected_ignores_other_replicas(orc, ki, t):
    k = _kernels(orc)[ki]
    clean, st0, _ = k.run(1, None)
    out, st1, det = k.run(1, _mk([_row(k, t, 1 + t[1] % 2)]))       # replica 1 or 2 does not exist
    assert all((a == b).all() for a, b in zip(out, clean)) and st1 == st0 and not det.any()
