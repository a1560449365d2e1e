#!/usr/bin/env python3
"""Generate tests/golden/* from the reference checkout.  Run ONLY where /root/reference exists (the build
container); the GPU box just reads the committed fixtures.

Sources (relative to /root/reference):
  mm     tests/mm_common/mm_tmr.c (side 9), tests/hifive1/matrixMultiply.tmr/mm.inc (19),
         tests/pynq/matrixMultiply.tmr/mm.inc (30): input matrices + xor_golden, parsed from the C initialisers;
         re-generated with mm_generator.py's algorithm (random.seed(0); randint(0, 2**32-1), tests/mm_common/
         mm_generator.py:42-51) to prove the generator equivalence, which then extends the set to side 32 and 256;
         result matrices come from the reference's own matrix_multiply (oracle/_ref).
  sha256 tests/sha256_common/sha_data.inc (LEN 10), tests/hifive1/sha256.tmr/sha_data.inc (LEN 4000).
  aes    tests/aes/ECB{GFSbox,KeySbox,VarKey,VarTxt}128.h: 568 NIST AESAVS vectors, 80 bytes each
         (key, key, ciphertext, plaintext, plaintext -- tests/aes/aes.c:62-66).
  crc16  no golden in the tree; vectors are outputs of the reference's crc16() (tests/crc16/crc16.c:21-31)
         compiled unmodified (oracle/_ref), incl. its own "Automated TMR" case (crc16.c:14,40).
"""
import hashlib
import json
import os
import random
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/tests"

from oracle import oracle as orc  # noqa: E402


def gen_mm(n, seed=0):
    """mm_generator.py:42-51 -- first matrix row-major, then second, from one seeded stream."""
    random.seed(seed)
    m1 = [[random.randint(0, 2**32 - 1) for _ in range(n)] for _ in range(n)]
    m2 = [[random.randint(0, 2**32 - 1) for _ in range(n)] for _ in range(n)]
    return np.array(m1, dtype=np.uint32), np.array(m2, dtype=np.uint32)


def parse_mm(path):
    txt = open(path).read()
    side = int(re.search(r"#define\s+side\s+(\d+)", txt).group(1))
    mats = []
    for name in ("first_matrix", "second_matrix"):
        body = re.search(name + r"\[\w+\]\[\w+\]\s*=\s*\{(.*?)\};", txt, re.S).group(1)
        vals = [int(v) for v in re.findall(r"\d+", body)]
        assert len(vals) == side * side, (name, len(vals))
        mats.append(np.array(vals, dtype=np.uint32).reshape(side, side))
    gold = int(re.search(r"xor_golden\s*=\s*(\d+)", txt).group(1))
    return side, mats[0], mats[1], gold


def parse_bytes(txt, name):
    body = re.search(name + r"\s*\[\w*\]\s*=\s*\{(.*?)\}", txt, re.S).group(1)
    return bytes(int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]+", body))


def main():
    orc.build()
    assert orc.ref() is not None, "oracle/_ref not built"
    out = {}

    # ---------------- mm
    mm = {}
    for path in ("mm_common/mm_tmr.c", "hifive1/matrixMultiply.tmr/mm.inc", "pynq/matrixMultiply.tmr/mm.inc"):
        side, f, s, gold = parse_mm(os.path.join(REF, path))
        gf, gs = gen_mm(side)
        assert (gf == f).all() and (gs == s).all(), "generator mismatch for side %d" % side
        r, err = orc.ref_mm(f, s, gold)
        assert err == 0
        mm["f%d" % side], mm["s%d" % side], mm["r%d" % side] = f, s, r
        out["mm_xor_golden_%d" % side] = gold
        out["mm_source_%d" % side] = path
    for side in (32, 256):
        f, s = gen_mm(side)
        gold = orc.mm_xor(orc.mm_plain(f, s))
        r, err = orc.ref_mm(f, s, gold)
        assert err == 0 and orc.mm_xor(r) == gold
        out["mm_xor_golden_%d" % side] = gold
        out["mm_inputs_sha256_%d" % side] = hashlib.sha256(f.tobytes() + s.tobytes()).hexdigest()
        out["mm_result_sha256_%d" % side] = hashlib.sha256(r.tobytes()).hexdigest()
        if side == 32:
            mm["f32"], mm["s32"], mm["r32"] = f, s, r
    # LANL variant (tests/matrixMultiply/matrixMultiply.c:80-81: both inputs i*j), run at side 32 through
    # the reference's mm_common matrix_multiply (identical loop nest, matrixMultiply.c:95-112)
    ij = np.fromfunction(lambda i, j: i * j, (32, 32), dtype=np.int64).astype(np.uint32)
    r, _ = orc.ref_mm(ij, ij, 0)
    mm["lanl_r32"] = r
    np.savez_compressed(os.path.join(HERE, "mm_fixtures.npz"), **mm)

    # ---------------- sha256
    sha = {}
    for tag, path in (("10", "sha256_common/sha_data.inc"), ("4000", "hifive1/sha256.tmr/sha_data.inc")):
        txt = open(os.path.join(REF, path)).read()
        data, gold = parse_bytes(txt, "hash_data"), parse_bytes(txt, "golden")
        assert len(data) == int(tag) and len(gold) == 32
        random.seed(0)  # sha_generator.py:44-50
        assert bytes(random.randint(0, 255) for _ in range(len(data))) == data
        assert hashlib.sha256(data).digest() == gold and orc.ref_sha256(data) == gold
        sha["data" + tag] = np.frombuffer(data, np.uint8)
        sha["golden" + tag] = np.frombuffer(gold, np.uint8)
    np.savez_compressed(os.path.join(HERE, "sha_fixtures.npz"), **sha)

    # ---------------- aes
    kat = []
    counts = {}
    for name in ("ECBGFSbox128", "ECBKeySbox128", "ECBVarKey128", "ECBVarTxt128"):
        txt = open(os.path.join(REF, "aes", name + ".h")).read()
        cnt = int(re.search(name + r"_count\s*=\s*(\d+)", txt).group(1))
        body = re.search(name + r"\[\]\s*=\s*\{(.*?)\};", txt, re.S).group(1)
        vals = bytes(int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]+", body))
        assert len(vals) >= cnt * 80, (name, len(vals), cnt)
        kat.append(np.frombuffer(vals[: cnt * 80], np.uint8).reshape(cnt, 80))
        counts[name] = cnt
    kat = np.concatenate(kat)
    assert kat.shape == (568, 80)
    for row in kat[::37]:  # spot check through the reference's aes_enc_dec
        key, key2, ct, pt, inp = (row[16 * q:16 * q + 16].tobytes() for q in range(5))
        enc, _ = orc.ref_aes(inp, key, 0)
        dec, _ = orc.ref_aes(enc, key2, 1)
        assert enc == ct and dec == pt
    np.savez_compressed(os.path.join(HERE, "aes_kat.npz"), kat=kat)
    out["aes_counts"] = counts
    sb = np.ctypeslib.as_array(orc.lib().orc_aes_sbox(), (256,))
    import ctypes as C
    rsb = (C.c_ubyte * 256).in_dll(orc.ref(), "ref_aes_rsbox")
    rs = (C.c_ubyte * 256).in_dll(orc.ref(), "ref_aes_sbox")
    assert bytes(rs) == sb.tobytes(), "oracle S-box differs from the reference's"
    assert bytes(rsb) == np.ctypeslib.as_array(orc.lib().orc_aes_rsbox(), (256,)).tobytes()
    out["aes_sbox_sha256"] = hashlib.sha256(sb.tobytes()).hexdigest()

    # ---------------- crc16 (outputs of the reference run here)
    vec = [{"data": b"Automated TMR".hex(), "crc": orc.ref_crc16(b"Automated TMR")}]
    assert vec[0]["crc"] == 0x5BA3
    rng = random.Random(16)
    for ln in list(range(0, 20)) + [63, 64, 65, 127, 128, 129, 200, 254, 255]:
        d = bytes(rng.randrange(256) for _ in range(ln))
        vec.append({"data": d.hex(), "crc": orc.ref_crc16(d)})
    vec.append({"data": b"123456789".hex(), "crc": orc.ref_crc16(b"123456789")})
    out["crc16_vectors"] = vec

    # ---------------- CHStone sha (tests/chstone/sha): the benchmark's own vectors, its expected digest (sha_driver.c:49-50)
    # and outputs of the reference run here on random inputs
    indata, stream_digest = orc.ref_chsha_vectors()
    txt = open(os.path.join(REF, "chstone/sha/sha_driver.c")).read()
    out_data = [int(v, 16) for v in re.findall(r"0x([0-9a-fA-F]{8})UL", txt)]
    assert len(out_data) == 5 and stream_digest.tolist() == out_data, "reference sha_stream() != its own outData"
    assert orc.ref_chsha(indata.tobytes()).tolist() == out_data
    rng = random.Random(5)
    chv = {"indata": indata, "outData": np.array(out_data, dtype=np.uint32)}
    for q, ln in enumerate((0, 64, 128, 192, 1024, 4096)):
        d = bytes(rng.randrange(256) for _ in range(ln))
        chv["rand%d" % q] = np.frombuffer(d, np.uint8)
        chv["rand%d_digest" % q] = orc.ref_chsha(d)
    np.savez_compressed(os.path.join(HERE, "chsha_fixtures.npz"), **chv)
    out["chsha_outData"] = out_data

    # ---------------- quicksort (tests/quicksort): the unmodified benchmark compiled natively; its main() never returns, so the
    # fixture is its stdout up to and including the second acknowledge line.  The E: blocks in it are the benchmark's own:
    # quick_sort_rev is an empty TODO (quicksort.c:131-133), so sub-tests 2 and 3 compare the sorted array with an unsorted golden.
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "qs_native")
        subprocess.check_call(["gcc", "-O0", "-w", os.path.join(REF, "quicksort/quicksort.c"), "-o", exe])
        p = subprocess.Popen(["stdbuf", "-oL", exe], stdout=subprocess.PIPE)
        buf = b""
        while True:
            line = p.stdout.readline()
            buf += line
            if line.startswith(b"# 100,"):
                break
        p.kill()
    out["quicksort_driver"] = {"acks": [ln.decode() for ln in buf.split(b"\r\n") if ln.startswith(b"#")],
                               "bytes_until_ack100": len(buf), "sha256_until_ack100": hashlib.sha256(buf).hexdigest()}
    # the reference sort on the benchmark's own first input (srand(0), 580 x rand()), as a checksum, plus random vectors
    qv = {}
    rng = random.Random(580)
    for q, n in enumerate((1, 2, 3, 17, 100, 580)):
        a = np.array([rng.randrange(-2**31, 2**31) for _ in range(n)], dtype=np.int32)
        qv["in%d" % q] = a
        qv["out%d" % q] = orc.ref_quicksort(a)
    np.savez_compressed(os.path.join(HERE, "quicksort_fixtures.npz"), **qv)

    # ---------------- CHStone aes (tests/chstone/aes): the reference's encrypt / decrypt through oracle/_ref's shim, all nine
    # Rijndael sizes, random blocks and keys; plus the benchmark's own vector (FIPS-197 Appendix B, aes.c:95-127)
    cav = {}
    rng = random.Random(197)
    for t in orc.CHAES_TYPES:
        nk, nb, _ = orc.chaes_geom(t)
        st = np.array([[rng.randrange(256) for _ in range(4 * nb)] for _ in range(12)], dtype=np.uint8)
        ky = np.array([[rng.randrange(256) for _ in range(4 * nk)] for _ in range(12)], dtype=np.uint8)
        if t == 128128:
            st[0] = [50, 67, 246, 168, 136, 90, 48, 141, 49, 49, 152, 162, 224, 55, 7, 52]
            ky[0] = [43, 126, 21, 22, 40, 174, 210, 166, 171, 247, 21, 136, 9, 207, 79, 60]
        cav["st%d" % t] = st
        cav["key%d" % t] = ky
        cav["enc%d" % t] = np.stack([orc.ref_chaes(st[q], ky[q], t, 0) for q in range(12)])
        cav["dec%d" % t] = np.stack([orc.ref_chaes(st[q], ky[q], t, 1) for q in range(12)])
    np.savez_compressed(os.path.join(HERE, "chaes_fixtures.npz"), **cav)

    # ---------------- crazyCF (tests/crazyCF, the CFCSS test program): the unmodified program compiled natively -- its two
    # output lines -- and, through oracle/_ref's shim, a grid of (srand argument, size) with timesThroughWhile at its 10
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "crazycf_native")
        subprocess.check_call(["gcc", "-O0", "-w", os.path.join(REF, "crazyCF/crazyCF.c"), "-o", exe])
        out["crazycf_stdout"] = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    grid = []
    rng = random.Random(42)
    for seed, size in [(42, 20), (1, 1), (2, 5), (3, 6), (4, 17), (5, 18), (6, 25), (7, 26), (8, 37), (9, 38), (10, 39), (0, 64),
                       (0x7FFFFFFF, 100), (0x80000000, 100), (0xFFFFFFFF, 50)] + [(rng.randrange(1 << 32), rng.randrange(1, 300))
                                                                                 for _ in range(49)]:
        grid.append([seed, size] + list(orc.ref_crazycf(seed, size)))
    out["crazycf_grid"] = grid  # rows: srand argument, size, Total, "total so far" value, number of such lines

    with open(os.path.join(HERE, "golden.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
