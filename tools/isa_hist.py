#!/usr/bin/env python3
"""tools/isa_hist.py [kernel-name-pattern] [--src FILE] [--extra "-DX=1 ..."] [--top N]

Static instruction histogram of the gfx950 code object of a kernel: how many instructions of each class a wave meets when it walks the kernel's text once
(loops are counted once: the children kernel's pass and the fc0 super-step are fully unrolled bodies inside one loop, so their counts are per child / per
super-step pair; read loop trip counts beside it).  Classes: MFMA by shape, VALU by family (convert / split, max / mul (LeakyReLU, scaling), address + select,
other), SALU, LDS, VMEM loads / stores, s_waitcnt, s_nop.  Used to put k_sib_children2 on an instruction diet and to count the MFMAs
behind bench.py's `executed_flops`.

Device-only compile of the source (hipcc --cuda-device-only), llvm-objdump -d of the unbundled code object."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

CLASSES = [
    ("mfma", re.compile(r"^v_mfma|^v_smfmac")),
    ("cvt/split", re.compile(r"^v_cvt|^v_fma_mix|^v_pack|^v_perm")),
    ("max/mul (lrelu, scale)", re.compile(r"^v_max|^v_min|^v_mul_f|^v_pk_mul|^v_pk_max|^v_pk_fma|^v_fma_f|^v_fmac|^v_pk_add_f|^v_add_f|^v_sub_f|^v_mad_f|^v_ldexp|^v_med3")),
    ("addr/select", re.compile(r"^v_add_u|^v_add_co|^v_addc|^v_sub_u|^v_sub_co|^v_lshl|^v_lshr|^v_ashr|^v_and|^v_or|^v_xor|^v_cndmask|^v_mov|^v_mad_u|^v_mad_i|^v_mul_u|^v_mul_i|^v_mul_lo|^v_mul_hi|^v_bfe|^v_bfi|^v_add3|^v_add_lshl|^v_lshl_add|^v_lshl_or|^v_and_or|^v_or3|^v_add_nc|^v_sub_nc|^v_accvgpr|^v_readlane|^v_readfirstlane|^v_writelane|^v_cmp|^v_alignbit|^v_mbcnt|^v_bcnt|^v_not")),
    ("valu other", re.compile(r"^v_")),
    ("lds", re.compile(r"^ds_")),
    ("vmem load", re.compile(r"^(global|buffer|flat|scratch)_load")),
    ("vmem store", re.compile(r"^(global|buffer|flat|scratch)_(store|atomic)")),
    ("s_waitcnt", re.compile(r"^s_waitcnt")),
    ("s_nop", re.compile(r"^s_nop")),
    ("s_barrier", re.compile(r"^s_barrier")),
    ("salu/branch", re.compile(r"^s_")),
]


def disassemble(src, extra, contract_off):
    import hashlib
    key = hashlib.sha256((open(src).read() + extra + str(contract_off)).encode()).hexdigest()[:16]  # (header edits: pass --extra " " to force)
    cache = os.path.join(tempfile.gettempdir(), f"isa_hist_{key}.s")
    if os.path.exists(cache):
        return open(cache).read()
    text = _disassemble(src, extra, contract_off)
    open(cache, "w").write(text)
    return text


def _disassemble(src, extra, contract_off):
    tmp = tempfile.mkdtemp(prefix="isa_hist_")
    dev, co = os.path.join(tmp, "dev.o"), os.path.join(tmp, "k.co")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fhip-fp32-correctly-rounded-divide-sqrt", "--cuda-device-only", "-w"]
    if contract_off:
        cmd.append("-ffp-contract=off")
    cmd += extra.split() + ["-c", src, "-I" + os.path.dirname(src), "-o", dev]
    subprocess.check_call(cmd)
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + dev,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def histogram(text):
    kernels, cur = collections.OrderedDict(), None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            kernels[cur] = collections.Counter()
            continue
        if cur is None:
            continue
        ins = line.strip().split()
        if not ins or ins[0].endswith(":"):
            continue
        kernels[cur][ins[0]] += 1
    return kernels


def classify(counter):
    by, detail = collections.Counter(), collections.defaultdict(collections.Counter)
    for op, c in counter.items():
        for name, rx in CLASSES:
            if rx.match(op):
                by[name] += c
                detail[name][op] += c
                break
        else:
            by["other"] += c
            detail["other"][op] += c
    return by, detail


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pattern", nargs="?", default="k_sib_children2")
    ap.add_argument("--src", default=os.path.join(ROOT, "omok-ai_amd", "csrc", "net_kernels.hip"))
    ap.add_argument("--extra", default=os.environ.get("EXTRA", ""))
    ap.add_argument("--top", type=int, default=8)
    ap.add_argument("--contract-off", action="store_true", help="tree_kernels.hip is built with -ffp-contract=off")
    ap.add_argument("--mfma", action="store_true", help="one line per kernel: its MFMA instructions by shape")
    a = ap.parse_args()
    kernels = histogram(disassemble(a.src, a.extra, a.contract_off))
    names = demangle(list(kernels))
    for k, cnt in kernels.items():
        pretty = names.get(k, k)
        if a.pattern not in pretty or not cnt:
            continue
        by, detail = classify(cnt)
        if a.mfma:
            print(f"{pretty[:110]:110s} " + ", ".join(f"{op} {c}" for op, c in detail["mfma"].most_common()))
            continue
        total = sum(by.values())
        valu = sum(by[c] for c in ("cvt/split", "max/mul (lrelu, scale)", "addr/select", "valu other"))
        print(f"== {pretty[:150]}")
        print(f"   instructions {total}: mfma {by['mfma']}, non-MFMA VALU {valu}, lds {by['lds']}, vmem load {by['vmem load']}, vmem store {by['vmem store']}, "
              f"s_waitcnt {by['s_waitcnt']}, s_nop {by['s_nop']}, s_barrier {by['s_barrier']}, salu/branch {by['salu/branch']}")
        for cname, _ in CLASSES:
            if by[cname]:
                tops = ", ".join(f"{op} {c}" for op, c in detail[cname].most_common(a.top))
                print(f"   {cname:24s} {by[cname]:6d}   {tops}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
