"""CPU checks of the compiled hot kernels and of the evidence that bench.py cites -- no GPU needed (hipcc cross-compiles gfx950).

Register allocation decides the speed of the matrix-core kernels: a spilled VGPR inside the MFMA loop comes back through scratch and
its `s_waitcnt vmcnt(0)` drains the operand prefetch (variants of mm_mfma_blk2_kernel with 35-117 spilled registers ran 7.9-13.4 ms
instead of 6.6, profiles/r02d_mm_blk2_ab_spills.txt).  These tests compile the kernels alone and hold the budget: registers, spills,
scratch, the number of MFMAs the step bodies issue, occupancy of the LDS-table kernels.  They also hold the bench's pointers: every
profiles/ file that traffic.json or bench.py names exists, every kernel name bench.py reports is a kernel in the sources."""
import json
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "coast_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

TU = r'''
#include "coast_hip.h"
#include <hip/hip_runtime.h>
#include "xmr.hpp"
#include "injector.hip"
#include "mm_kernel.hip"
#include "mm_mfma_kernel.hip"
#include "mm_mfma_blk2_kernel.hip"
#include "mm_mfma_blk3_kernel.hip"
#include "mm_mfma_blk4_kernel.hip"
#include "sha256_kernel.hip"
#include "aes_kernel.hip"
#include "crc16_kernel.hip"
namespace coast {
#define MMARGS const uint32_t *, const uint32_t *, uint32_t *, uint32_t, Counters, FaultTab, uint8_t *
template __global__ void mm_mfma_blk3_kernel<3, true>(MMARGS);
template __global__ void mm_mfma_blk3_kernel<3, false>(MMARGS);
template __global__ void mm_mfma_blk3_kernel<3, false, 0, true>(MMARGS);
template __global__ void mm_mfma_blk2_kernel<3, true>(MMARGS);
template __global__ void mm_mfma_blk2_kernel<3, false>(MMARGS);
template __global__ void mm_mfma_blk4_kernel<false, false>(MMARGS);
template __global__ void mm_mfma_blk4_kernel<false, true>(MMARGS);
#define AESARGS uint8_t *, uint8_t *, uint64_t, uint64_t, Counters, FaultTab, uint8_t *, size_t
template __global__ void aes128_enc_rep_kernel<2>(AESARGS);
template __global__ void aes128_dec_rep_kernel<2>(AESARGS);
template __global__ void aes128_enc_fast_kernel<3>(AESARGS);
template __global__ void aes128_dec_fast_kernel<3>(AESARGS);
template __global__ void sha256_fast_kernel<3, true>(const uint8_t *, size_t, uint32_t, uint64_t, uint8_t *, uint64_t, Counters, FaultTab, uint8_t *, size_t, size_t);
#define CRCARGS const uint8_t *, uint32_t, uint64_t, uint16_t *, const uint16_t *, uint64_t, uint64_t, Counters, FaultTab, uint8_t *, size_t, size_t
template __global__ void crc16_stream_kernel<3, 2, true>(CRCARGS);
template __global__ void crc16_stream_kernel<3, 1, false>(CRCARGS);
}
'''


@pytest.fixture(scope="module")
def compiled(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    d = tmp_path_factory.mktemp("budget")
    src = str(d / "budget_tu.hip")  # outside the source tree: coast_amd/csrc is content-hashed into the library
    with open(src, "w") as fh:
        fh.write(TU)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-I", CSRC, "-I", os.path.join(ROOT, "include")]
    rem = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", str(d / "t.o")], capture_output=True,
                         text=True, timeout=900)
    assert rem.returncode == 0, rem.stderr[-3000:]
    asm = d / "t.s"
    subprocess.check_call(cmd + ["-S", src, "-o", str(asm)], stderr=subprocess.DEVNULL, timeout=900)
    usage = {}
    cur = None
    for line in rem.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = usage.setdefault(subprocess.check_output(["c++filt", m.group(1)], text=True).strip(), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    text = asm.read_text()
    bodies = {}
    for m in re.finditer(r"^(_ZN5coast\w+):[^\n]*\n(.*?)\n\s+s_endpgm", text, flags=re.S | re.M):
        bodies[subprocess.check_output(["c++filt", m.group(1)], text=True).strip()] = m.group(2)
    return usage, bodies


def _find(table, prefix):
    hits = [k for k in table if k.startswith(prefix)]
    assert len(hits) == 1, (prefix, sorted(table))
    return table[hits[0]]


@pytest.mark.parametrize("flags", ["true", "false"])
def test_mm_register_block_kernels_fit_their_register_files(compiled, flags):
    usage, bodies = compiled
    # two waves per SIMD: 256 VGPRs, no AGPRs; the default TMR kernel (blk3) and its predecessor (blk2)
    # per-kernel limits: blk2 at its round-3 values, blk3 at the ones measured with it
    for kern, spill_cap, hot_loads in (("mm_mfma_blk3_kernel", 10, 2), ("mm_mfma_blk2_kernel", 8, 1)):
        tail = ", 0, false>" if kern == "mm_mfma_blk3_kernel" else ">"  # (blk3's third / fourth parameters: physical-upset hooks, cloned staging)
        u2 = _find(usage, "void coast::%s<3, %s%s" % (kern, flags, tail))
        assert u2["VGPRs"] <= 256 and u2["AGPRs"] == 0 and u2["Occupancy [waves/SIMD]"] == 2
        # the armed-upset hook (cold, wave-uniform branch) may park a few registers; the step bodies must not
        assert u2["VGPRs Spill"] <= spill_cap, (kern, u2)
        b2 = _find(bodies, "void coast::%s<3, %s%s" % (kern, flags, tail))
        # 4 step variants x 60 MFMAs x 2 row halves: every k-slab of a tile issues its 120 MFMAs once
        assert len(re.findall(r"v_mfma_i32_16x16x64_i8", b2)) == 480
        assert len(re.findall(r"scratch_load", b2)) <= 12
        # ... and inside the tile loops (the basic blocks that hold the MFMAs) at most the one reload per tile of a tally register
        for blk in re.split(r"\n\.LBB\d+_\d+:", b2):
            if len(re.findall(r"v_mfma_i32_16x16x64_i8", blk)) >= 60:
                assert len(re.findall(r"scratch_load", blk)) <= hot_loads and not re.search(r"scratch_store", blk), kern
    # blk3: every set of ten MFMAs reads its own four A fragments (24 + 12 B fragment reads per step and wave)
    b3 = _find(bodies, "void coast::mm_mfma_blk3_kernel<3, %s, 0, false>" % flags)
    assert len(re.findall(r"ds_read_b128", b3)) >= 8 * 36
    if flags == "false":
        # COAST_F_CLONE_STAGING: the register file is full -- the twelve clone registers must not cost a reload inside the MFMA blocks (a
        # scratch reload drains the VMEM queue: builds that had them ran + 30 % instead of + 15 %, profiles/r05_mm_clone_ab.txt).  Held here:
        # two waves per SIMD, the MFMA count, the clone loads, no scratch traffic in the blocks that hold MFMAs
        uc = _find(usage, "void coast::mm_mfma_blk3_kernel<3, false, 0, true>")
        bc = _find(bodies, "void coast::mm_mfma_blk3_kernel<3, false, 0, true>")
        assert uc["VGPRs"] <= 256 and uc["Occupancy [waves/SIMD]"] == 2 and len(re.findall(r"v_mfma_i32_16x16x64_i8", bc)) == 480
        # eight more two-word loads per converted slab (one slab per wave and two steps: the off-duty variants of the 4 x 2 step bodies)
        assert len(re.findall(r"buffer_load_dwordx2", bc)) - len(re.findall(r"buffer_load_dwordx2", b3)) >= 8 * 4
        for blk in re.split(r"\n\.LBB\d+_\d+:", bc):
            if len(re.findall(r"v_mfma_i32_16x16x64_i8", blk)) >= 10:
                assert not re.search(r"scratch_(load|store)", blk)


@pytest.mark.parametrize("clone", ["true", "false"])
def test_mm_128_row_panel_kernel_fits_its_register_file(compiled, clone):
    """mm_mfma_blk4_kernel (round 6, the TMR default): two waves per SIMD, no register in scratch at all, twelve step bodies of sixty MFMAs in
    straight-line code and one clean inner loop (two alternative bodies that join, or an inner loop with a mid-loop exit, made the register
    allocator reload accumulator tuples in front of MFMAs: hundreds of spills), every fragment read's address one register + an instruction
    offset (the f panel as [row][plane][256 B])."""
    usage, bodies = compiled
    u = _find(usage, "void coast::mm_mfma_blk4_kernel<false, %s, 0>" % clone)
    b = _find(bodies, "void coast::mm_mfma_blk4_kernel<false, %s, 0>" % clone)
    assert u["VGPRs"] <= 256 and u["AGPRs"] == 0 and u["Occupancy [waves/SIMD]"] == 2 and u["VGPRs Spill"] == 0 and u["ScratchSize [bytes/lane]"] == 0, u
    assert len(re.findall(r"v_mfma_i32_16x16x64_i8", b)) == 12 * 60 and not re.search(r"scratch_(load|store)", b)
    # 24 A + 12 B fragment reads per step body; a step body without tile end and f work issues at most 40 VALU instructions beside its MFMAs
    assert len(re.findall(r"ds_read_b128", b)) >= 12 * 36
    lines = [ln.strip() for ln in b.split("\n") if ln.strip() and not ln.strip().startswith((";", ".")) and not ln.split(";")[0].strip().endswith(":")]
    idx = [k for k, ln in enumerate(lines) if ln.startswith("v_mfma")]
    lean = min(sum(1 for ln in lines[idx[60 * k]:idx[60 * k + 59]] if ln.startswith("v_") and not ln.startswith(("v_mfma", "v_readlane", "v_writelane")))
               for k in range(12))
    assert lean <= (40 if clone == "false" else 60), lean
    if clone == "true":  # the clones: four more one-word loads of s per step body, two four-word loads of f more per f piece
        b0 = _find(bodies, "void coast::mm_mfma_blk4_kernel<false, false, 0>")
        assert len(re.findall(r"buffer_load_dword ", b)) - len(re.findall(r"buffer_load_dword ", b0)) >= 12 * 4
        assert len(re.findall(r"buffer_load_dwordx4", b)) > len(re.findall(r"buffer_load_dwordx4", b0))


def test_lds_table_kernels_keep_their_occupancy(compiled):
    usage, bodies = compiled
    enc = _find(usage, "void coast::aes128_enc_rep_kernel<2>")
    dec = _find(usage, "void coast::aes128_dec_rep_kernel<2>")
    # encryption: two 1024-thread workgroups per CU (64 KiB of tables each) need <= 64 registers per lane
    assert enc["VGPRs"] <= 64 and enc["VGPRs Spill"] == 0 and enc["Occupancy [waves/SIMD]"] == 8
    assert dec["VGPRs"] <= 128 and dec["VGPRs Spill"] == 0
    # lookups per block: encryption 203 dword reads; decryption 208 dword + 32 eight-byte pair reads (bench.py AES.LOOKUPS).  The
    # rounds are instantiated twice -- clean tiles, and tiles that own an armed upset (injector hooks between the same rounds)
    be, bd = _find(bodies, "void coast::aes128_enc_rep_kernel<2>"), _find(bodies, "void coast::aes128_dec_rep_kernel<2>")
    # (the compiler may sink a few lookups of the last round below the join of the two instantiations)
    ne, nd, nd64 = len(re.findall(r"ds_read_b32", be)), len(re.findall(r"ds_read_b32", bd)), len(re.findall(r"ds_read_b64", bd))
    # (eight-byte reads of the encryption kernel: the armed tiles' per-round look at the lane's upset records, 11 hook points)
    assert 2 * 203 - 8 <= ne <= 2 * 203 and len(re.findall(r"ds_read_b64", be)) <= 11, ne
    assert 2 * 208 - 8 <= nd <= 2 * 208 and 2 * 24 <= nd64 <= 2 * 32 + 11, (nd, nd64)  # (pair reads whose Tis half is dead are narrowed to dwords)
    for name in ("void coast::crc16_stream_kernel<3, 2, true, 1024,", "void coast::crc16_stream_kernel<3, 1, false, 1024,"):
        u = _find(usage, name)  # 1024-thread persistent workgroups: 128 registers per lane
        assert u["VGPRs"] <= 128 and u["VGPRs Spill"] == 0, (name, u)


def test_kernels_that_address_lds_from_zero_have_no_static_lds(compiled):
    """the aes rep kernels (and the matrix-core kernels) build LDS addresses by permuting bytes / from constants that assume
    the dynamic segment starts at LDS offset 0; they guard it with a run-time trap.  A static `__shared__` added to one of them would
    move the dynamic segment up -- this holds the layout at build time so that the trap can never be what reports it."""
    usage, _ = compiled
    for name in ("aes128_enc_rep_kernel<2>", "aes128_dec_rep_kernel<2>", "mm_mfma_blk3_kernel<3, true, 0, false>",
                 "mm_mfma_blk3_kernel<3, false, 0, false>", "mm_mfma_blk3_kernel<3, false, 0, true>", "mm_mfma_blk2_kernel<3, true>", "mm_mfma_blk2_kernel<3, false>"):
        assert _find(usage, "void coast::" + name)["LDS Size [bytes/block]"] == 0, name


def test_injector_hooks_cost_the_lean_kernels_nothing(compiled):
    """round 3: the lean sha256 / aes / crc16 kernels carry their own injector hooks (a tile that owns an armed upset takes a
    wave-uniform branch).  The hooks must not put the kernels on scratch, and must not take occupancy from the clean path."""
    usage, _ = compiled
    sha = _find(usage, "void coast::sha256_fast_kernel<3, true>")
    assert sha["VGPRs"] <= 64 and sha["Occupancy [waves/SIMD]"] == 8 and sha["ScratchSize [bytes/lane]"] == 0, sha
    for name in ("aes128_enc_fast_kernel<3>", "aes128_dec_fast_kernel<3>", "aes128_enc_rep_kernel<2>", "aes128_dec_rep_kernel<2>",
                 "crc16_stream_kernel<3, 2, true, 1024,", "crc16_stream_kernel<3, 1, false, 1024,"):
        u = _find(usage, "void coast::" + name)
        assert u["ScratchSize [bytes/lane]"] == 0 and u["VGPRs Spill"] == 0, (name, u)
    assert _find(usage, "void coast::aes128_enc_fast_kernel<3>")["VGPRs"] <= 64


def test_in_kernel_counter_fold_orders_by_atomics_not_by_an_l2_write_back(compiled):
    """round 5, block_fold (xmr.hpp; the default since round 6, COAST_AES_FOLD=0 takes it out): the last workgroup out of a persistent aes kernel folds the counter slots.  Its
    ordering is a returning atomic on the slot's own line in front of the ticket -- a `__threadfence()` there is `buffer_wbl2` on gfx950,
    a write-back of an L2 full of the launch's own output: + 8 us on a 60 us kernel (profiles/r05_aes_step.txt).  Holds that at build
    time: the fold is compiled into both kernels (three read-and-clear exchanges), and no L2 write-back is."""
    _, bodies = compiled
    for name in ("void coast::aes128_enc_rep_kernel<2>", "void coast::aes128_dec_rep_kernel<2>"):
        b = _find(bodies, name)
        # three read-and-clear exchanges of the fold + the returning exchange on the slot line's pad word (round 6: an exchange with 0 -- the
        # pad stays 0 -- where round 5 added 1)
        assert len(re.findall(r"global_atomic_swap_x2", b)) == 4, name
        assert "buffer_wbl2" not in b and "buffer_inv" not in b, name
        # the slot line's returning exchange (sc0) sits in front of the ticket's returning add, an s_waitcnt vmcnt(0) between them
        i_line, i_ticket = b.find("global_atomic_swap_x2 v["), b.find("global_atomic_add v")
        assert 0 < i_line < i_ticket and "sc0" in b[i_line:b.find("\n", i_line)] and "sc0" in b[i_ticket:b.find("\n", i_ticket)], name
        assert "s_waitcnt vmcnt(0)" in b[i_line:i_ticket], name


def test_counter_fold_ticket_is_two_level(compiled):
    """round 6, block_fold's completion count (xmr.hpp): one returning add on the workgroup's GROUP word, and only behind it -- for the group's
    last workgroup -- the reset of that word and the returning add on the top word.  One word for every workgroup cost 2.7 / 5.2 us per launch
    (same-address device-scope atomics retire at ~10 ns each: profiles/r06_aes_two_level_ticket.txt).  The arithmetic the kernel uses to find a
    group's size must count every workgroup exactly once for any grid, and the host must allocate the words the kernel addresses."""
    src = open(os.path.join(CSRC, "xmr.hpp")).read()
    groups = int(re.search(r"constexpr uint32_t kTicketGroups = (\d+);", src).group(1))
    stride = int(re.search(r"constexpr uint32_t kTicketStride = (\d+);", src).group(1))
    assert "kTicketWords = (1 + kTicketGroups) * kTicketStride" in src and stride * 4 >= 128  # a 128-byte line per ticket
    assert "(gridDim.x - g + kTicketGroups - 1u) / kTicketGroups" in src and "gridDim.x < kTicketGroups ? gridDim.x : kTicketGroups" in src
    for grid in list(range(1, 70)) + [255, 256, 257, 300, 511, 512, 513, 1024]:
        members = [(grid - g + groups - 1) // groups for g in range(min(groups, grid))]
        assert all(m >= 1 for m in members) and sum(members) == grid, grid
        assert members == [len(range(g, grid, groups)) for g in range(min(groups, grid))], grid
    host = open(os.path.join(CSRC, "coast_hip.hip")).read()
    assert "sizeof(uint32_t) * (size_t)kTicketWords" in host  # kSlotBytes: slots + ticket words, cleared together at create / reset
    _, bodies = compiled
    for name in ("void coast::aes128_enc_rep_kernel<2>", "void coast::aes128_dec_rep_kernel<2>"):
        b = _find(bodies, name)
        adds = [m.start() for m in re.finditer(r"global_atomic_add v\d+, .* sc0", b)]
        swap = b.find("global_atomic_swap v")
        assert len(adds) == 2 and adds[0] < swap < adds[1], (name, adds, swap)  # group ticket, the group word's reset, top ticket


def test_bench_lookup_counts_match_the_compiled_kernels():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.AES.LOOKUPS == {0: 203, 1: 208 + 32}


def test_evidence_pointers_resolve():
    """every profiles/ file that traffic.json or bench.py cites is tracked here, every kernel bench.py names is in the sources"""
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    cited = set()
    for rec in traffic.values():
        cited.update(re.findall(r"profiles/[\w./-]+\.(?:txt|json)", rec.get("source", "")))
    bench = open(os.path.join(ROOT, "bench.py")).read()
    cited.update(re.findall(r"profiles/[\w./-]+\.(?:txt|json)", bench))
    for f in sorted(cited):
        assert os.path.exists(os.path.join(ROOT, f)), "cited evidence %s is missing" % f
    sources = "".join(open(os.path.join(CSRC, f)).read() for f in os.listdir(CSRC) if f.endswith((".hip", ".inc")))
    for kern in set(re.findall(r"\b((?:mm|aes128|crc16|sha256|chsha|cache_test)_\w*kernel)\b", bench)):
        assert re.search(r"\b%s\b" % kern, sources), "bench.py names %s, which is not a kernel in coast_amd/csrc" % kern
