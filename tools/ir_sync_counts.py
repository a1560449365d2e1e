#!/usr/bin/env python3
"""Count, on the reference's own -O0 LLVM IR, the sync points that COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC add to a function: every
EXECUTED conditional branch (`br i1`: syncTerminator, synchronization.cpp:146-155, 741-949) and every executed getelementptr whose last
index is not a constant (syncGEP votes the last operand, :413-474; constant offsets return early, :428-431), split by what the address
feeds (a load: -noLoadSync drops it; a store: -noStoreAddrSync).  The IR is what `clang -O0 -emit-llvm` makes of the C file where it lies
under /root/reference (the reference's flow compiles at -O0 before opt, tests/makefiles/Makefile.common); the instrumentation is a call
in front of each such instruction of the chosen functions, the counts come from RUNNING the instrumented code on the benchmark's own kind
of input.  tests/test_ir_counts_cpu.py compares them with the oracle's schedules (coast_oracle.c: mm_call_indexed, aes_item_indexed,
ct_item_indexed, chsha_item_indexed).  Needs the reference checkout: a container-side pin, like oracle/_ref.
CAVEAT: CLANG below is the image's LLVM 22; the pass runs on clang-7's -O0 IR (tests/makefiles/Makefile.compile:3-10) and LLVM 7 is not
available here.  The statement classes counted (one conditional branch per evaluated condition, one GEP per variable subscript, one store
per assignment at -O0) are what -O0 lowering has produced for C like this across those versions, but nothing here checks it."""
import os
import re
import subprocess
import sys
import tempfile

CLANG = "/opt/rocm/lib/llvm/bin/clang"
REF = "/root/reference/tests"


def _is_alloca(lines, func, store_line):
    """is the pointer operand of this store one of the function's allocas (a local variable at -O0)?"""
    ptr = re.search(r", ptr (%[\w.]+)", store_line).group(1)
    inside = False
    for ln in lines:
        m = re.match(r"define .*@([\w.]+)\(", ln)
        if m:
            inside = m.group(1) == func
        elif inside and ln.startswith("}"):
            return False
        elif inside and re.match(r"\s+%s = alloca " % re.escape(ptr), ln):
            return True
    return False


def instrument(ll, funcs):
    """returns the IR text with `call void @__coast_cnt(i32 k)` in front of the counted instructions of `funcs`
    (k = 0 conditional branch, 1 variable GEP feeding a load, 2 feeding a store, 3 feeding something else, 4 store of a computed value
    to memory the function does not own, 5 the same into one of its own allocas)"""
    out, cur, body = [], None, []
    lines = ll.split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"define .*@([\w.]+)\(", ln)
        if m:
            cur = m.group(1) if m.group(1) in funcs else None
        if cur and ln.startswith("}"):
            cur = None
        if cur:
            if re.match(r"\s+br i1 ", ln):
                out.append("  call void @__coast_cnt(i32 0)")
            # a switch votes its operand like a conditional branch does (syncTerminator, :761-767); a `ret` of a computed value votes it
            # (:755-760; `ret i32 0` has nothing cloned to vote)
            if re.match(r"\s+switch i\d+ %", ln):
                out.append("  call void @__coast_cnt(i32 6)")
            if re.match(r"\s+ret (?!void)\S+ %", ln):
                out.append("  call void @__coast_cnt(i32 7)")
            # a store of a computed, non-pointer value: a store-data sync point under -noMemReplication (:197-224) -- at -O0 that includes
            # the stores into the allocas of the function's own locals (reported apart: this design keeps locals in registers)
            if re.match(r"\s+store (?!ptr )\S+ %[\w.]+, ptr ", ln):
                out.append("  call void @__coast_cnt(i32 %d)" % (5 if re.match(r"\s+store \S+ %[\w.]+, ptr %\d+,", ln) and _is_alloca(lines, cur, ln) else 4))
            g = re.match(r"\s+(%[\w.]+) = getelementptr .*, \w+ (\S+)$", ln)
            if g and g.group(2).startswith("%"):
                res, kind = g.group(1), 3
                for nxt in lines[i + 1:i + 60]:  # the class of the address: its first user, through a GEP that feeds a GEP (:343-351)
                    if nxt.startswith("}"):
                        break
                    if re.search(r"= load .*ptr %s(,|$)" % re.escape(res), nxt):
                        kind = 1
                        break
                    if re.search(r"store .*, ptr %s(,|$)" % re.escape(res), nxt):
                        kind = 2
                        break
                    g2 = re.match(r"\s+(%%[\w.]+) = getelementptr .*ptr %s," % re.escape(res), nxt)
                    if g2:
                        res = g2.group(1)
                        continue
                    if re.search(r"%s\b" % re.escape(res), nxt):
                        break
                out.append(ln)
                out.append("  call void @__coast_cnt(i32 %d)" % kind)
                i += 1
                continue
        out.append(ln)
        i += 1
    return "\n".join(out) + "\ndeclare void @__coast_cnt(i32)\n"


DRIVER_HEAD = r'''
#include <stdio.h>
#include <string.h>
static unsigned long cnt[8];
static int frozen;
void __coast_cnt(int k) { if (!frozen) cnt[k]++; }
static void report(const char *tag) { fprintf(stderr, "%s %lu %lu %lu %lu %lu %lu %lu %lu\n", tag, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6], cnt[7]); for (int i = 0; i < 8; i++) cnt[i] = 0; frozen = 0; }
'''


def run_case(src, funcs, driver, cflags=(), rename=None, opt=None):
    """opt: an optimisation level to run over the -O0 IR first (`-O3` for the benchmarks whose Makefile puts it in front of the pass) --
    THIS toolchain's pipeline, not LLVM 7's: supporting evidence, not a pin"""
    with tempfile.TemporaryDirectory() as td:
        lls = []
        for n, one in enumerate([src] if isinstance(src, str) else list(src)):  # (a benchmark of several translation units: each its own IR)
            ll = os.path.join(td, "ref%d.ll" % n)
            subprocess.check_call([CLANG, "-O0", "-S", "-emit-llvm", "-w", *(["-Xclang", "-disable-O0-optnone"] if opt else []), *cflags, one, "-o", ll])
            if opt:
                subprocess.check_call([os.path.join(os.path.dirname(CLANG), "opt"), opt, "-S", ll, "-o", ll])
            text = open(ll).read()
            for a, b in (rename or {}).items():
                text = re.sub(r"@%s\b" % re.escape(a), "@" + b, text)
            text = instrument(text, funcs)
            if n:
                text = text.replace("\ndeclare void @__coast_cnt(i32)\n", "\n") if "declare void @__coast_cnt" in text.split("\n")[0] else text
            open(ll, "w").write(text)
            lls.append(ll)
        drv = os.path.join(td, "drv.c")
        open(drv, "w").write(DRIVER_HEAD + driver)
        exe = os.path.join(td, "run")
        subprocess.check_call([CLANG, "-O0", "-w", *lls, drv, "-o", exe])
        res = {}
        for line in subprocess.run([exe], text=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True).stderr.strip().split("\n"):
            tag, *v = line.split()
            res[tag] = {"branches": int(v[0]), "gep_loads": int(v[1]), "gep_stores": int(v[2]), "gep_other": int(v[3]),
                        "stores_to_memory": int(v[4]), "stores_to_local_allocas": int(v[5]), "switches": int(v[6]), "returns": int(v[7])}
        return res


def aes():
    drv = r'''
void aes_enc_dec(unsigned char *state, unsigned char *key, unsigned char dir);
int main(void) {
    unsigned char st[16], ky[16];
    for (int i = 0; i < 16; i++) st[i] = (unsigned char)(17 * i + 3), ky[i] = (unsigned char)(29 * i + 7);
    aes_enc_dec(st, ky, 0); report("aes_enc");
    aes_enc_dec(st, ky, 1); report("aes_dec");
    return 0; }
'''
    return run_case(os.path.join(REF, "aes", "TI_aes_128.c"), {"aes_enc_dec"}, drv, cflags=["-I" + os.path.join(REF, "aes")])


def mm(n=9):
    """tests/mm_common/mm.c (the side-9 program; it includes mm_common.c): matrix_multiply on its own matrices"""
    drv = r'''
extern void mm_run_test(void);
extern unsigned int first_matrix[9][9], second_matrix[9][9], results_matrix[9][9];
extern void matrix_multiply(unsigned int f[][9], unsigned int s[][9], unsigned int r[][9]);
int main(void) { matrix_multiply(first_matrix, second_matrix, results_matrix); report("mm"); return 0; }
'''
    assert n == 9
    return run_case(os.path.join(REF, "mm_common", "mm.c"), {"matrix_multiply"}, drv,
                    cflags=["-I" + os.path.join(REF, "mm_common"), "-I/root/reference/tests"], rename={"main": "ref_main"})


def cache_test(n=600, corrupt=()):
    drv = r'''
extern int calc_sum(int *array);
extern void generateGolden(void);
static int arr[%d];
int main(void) { generateGolden(); for (int i = 0; i < %d; i++) arr[i] = i; %s
    calc_sum(arr); report("calc_sum"); return 0; }
''' % (n, n, " ".join("arr[%d] = -5;" % k for k in corrupt))
    return run_case(os.path.join(REF, "cache_test", "cacheTest.c"), {"calc_sum"}, drv, rename={"main": "ref_main"})


def chsha(nbytes=128):
    drv = r'''
typedef unsigned char BYTE;
extern void sha_init(void); extern void sha_update(const BYTE *, int); extern void sha_final(void);
static BYTE buf[%d];
int in_i[2]; BYTE indata[2][8192]; /* sha_stream's inputs (sha.h): unused here, the driver calls sha_update itself */
int main(void) { for (int i = 0; i < %d; i++) buf[i] = (BYTE)(i * 31 + 5);
    sha_init(); sha_update(buf, %d); sha_final(); report("chsha"); return 0; }
''' % (max(nbytes, 1), nbytes, nbytes)
    return run_case(os.path.join(REF, "chstone", "sha", "sha.c"),
                    {"sha_transform", "sha_update", "sha_final"}, drv, cflags=["-I" + os.path.join(REF, "chstone", "sha")],
                    rename={"main": "ref_main", "memcpy": "ch_memcpy", "memset": "ch_memset"})  # (sha.c defines its own, :50-80)


def chaes(types=(128128, 192192, 256256, 128256, 256128)):
    """tests/chstone/aes: encrypt / decrypt of one block (KeySchedule, the rounds); counting stops at the function's first printf --
    the print-out and the comparison with the golden block behind it are outside what coast_chaes_batch computes"""
    drv = r'''
#include <stdarg.h>
int ch_printf(const char *f, ...) { (void)f; frozen = 1; return 0; }
extern int encrypt(int *, int *, int); extern int decrypt(int *, int *, int);
int main(void) { int st[32], ky[32];
%s return 0; }
''' % " ".join('for (int i = 0; i < 32; i++) st[i] = (i * 37 + 11) & 255, ky[i] = (i * 59 + 3) & 255; encrypt(st, ky, %d); report("enc_%d"); '
                  'decrypt(st, ky, %d); report("dec_%d");' % (t, t, t, t) for t in types)
    d = os.path.join(REF, "chstone", "aes")
    return run_case([os.path.join(d, f) for f in ("aes.c", "aes_enc.c", "aes_dec.c", "aes_key.c", "aes_func.c")], {"encrypt", "decrypt", "KeySchedule", "SubByte", "ByteSub_ShiftRow", "InversShiftRow_ByteSub",
                                                  "MixColumn_AddRoundKey", "AddRoundKey_InversMixColumn", "AddRoundKey"}, drv,
                    cflags=["-I" + d, "-include", "stdio.h"], rename={"main": "ref_main", "printf": "ch_printf"})


def crazycf():
    """tests/crazyCF/crazyCF.c under -TMR (unittest/cfg/full_tmr.yml:8): main() and fillArray() with the program's own constants
    (srand(42), size 20, timesThroughWhile 10)"""
    drv = r'''
#include <stdarg.h>
int ch_printf(const char *f, ...) { (void)f; return 0; }
extern int ref_main(void);
int main(void) { ref_main(); report("crazycf"); return 0; }
'''
    return run_case(os.path.join(REF, "crazyCF", "crazyCF.c"), {"ref_main", "fillArray"}, drv, rename={"main": "ref_main", "printf": "ch_printf"})


def crc16(lengths=(0, 13, 255)):
    drv = r'''
unsigned short crc16(const unsigned char *data_p, unsigned char length);
int main(void) { unsigned char d[255]; for (int i = 0; i < 255; i++) d[i] = (unsigned char)(i * 7 + 1);
%s return 0; }
''' % " ".join('crc16(d, %d); report("crc16_%d");' % (n, n) for n in lengths)
    return run_case(os.path.join(REF, "crc16", "crc16.c"), {"crc16"}, drv, cflags=["-I" + REF], rename={"main": "ref_main"})


def sha256(lengths=(0, 3, 64, 56, 119)):
    """tests/sha256_common (OPT_FLAGS empty: the -O0 shape; the hifive1 build of the same source runs -O3 in front of the pass)"""
    drv = r'''
void sha256_hash(unsigned char ctx_data[], unsigned ctx_bitlen[], unsigned ctx_state[], unsigned char data[], unsigned len, unsigned char hash[]);
int main(void) { unsigned char h[32], cd[64], m[192]; unsigned bl[2], st[8]; for (int i = 0; i < 192; i++) m[i] = (unsigned char)(i * 3 + 1);
%s return 0; }
''' % " ".join('sha256_hash(cd, bl, st, m, %d, h); report("sha256_%d");' % (n, n) for n in lengths)
    src = os.path.join(REF, "sha256_common", "sha256_tmr.c")
    cf = ["-I" + REF, "-I" + os.path.join(REF, "sha256_common")]
    out = run_case(src, {"sha256_hash", "sha256_transform"}, drv, cflags=cf, rename={"main": "ref_main"})
    for k, v in run_case(src, {"sha256_hash"}, drv, cflags=cf, rename={"main": "ref_main"}).items():
        out[k + "_hash_function_alone"] = v
    return out


def quicksort(values):
    """tests/quicksort (its Makefile runs -O3 in front of the pass): quick_sort on the given ints, IR after this toolchain's -O3"""
    drv = r'''
extern void quick_sort(int *A, int len);
static int a[%d] = {%s};
int main(void) { quick_sort(a, %d); report("quick_sort"); return 0; }
''' % (len(values), ",".join(str(int(v)) for v in values), len(values))
    return run_case(os.path.join(REF, "quicksort", "quicksort.c"), {"quick_sort"}, drv, cflags=["-I" + REF], rename={"main": "ref_main"},
                    opt="-O3")


if __name__ == "__main__":
    which = sys.argv[1:] or ["aes", "mm", "cache_test", "chsha", "crc16", "sha256"]
    for w in which:
        print(w, globals()[w]())
