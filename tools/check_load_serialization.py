#!/usr/bin/env python3
"""Static check of the gfx950 assembly of the HIP kernels for serialized loads: a buffer load, then
`s_waitcnt vmcnt(0)`, then another load with no consumer (MFMA / LDS store / barrier) in between.  That pattern means
every load of a stage pays a full memory latency (weight-gradient loaders).  No GPU needed.

    python tools/check_load_serialization.py [file.hip ...]     (default: the conv kernels)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "domain-transfer-gan_amd", "csrc")


def scan(asm):
    cur, last, report = None, None, {}
    for i, l in enumerate(asm.split("\n")):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, last = m.group(1), None
            continue
        if cur is None:
            continue
        if "buffer_load" in l:  # every loader uses buffer loads; the epilogues' bias loads are global_load
            if last is not None and last[1]:
                report[cur] = report.get(cur, 0) + 1
            last = [i, False]
        elif "s_waitcnt vmcnt(0)" in l and last is not None and i - last[0] < 40:
            last[1] = True
        elif ("v_mfma" in l or "ds_write" in l or "s_barrier" in l) and last is not None:
            last = None
        if "s_endpgm" in l:
            cur = None
    return report


def main():
    files = sys.argv[1:] or [os.path.join(CSRC, f) for f in ("conv_igemm.hip", "conv_wgrad.hip", "conv_bf16.hip", "conv_x3.hip")]
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for f in files:
            base = os.path.splitext(os.path.basename(f))[0]
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", f, "--save-temps=obj",
                                   "-o", os.path.join(td, base + ".o")], cwd=td, stderr=subprocess.DEVNULL)
            rep = scan(open(os.path.join(td, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read())
            for k, v in sorted(rep.items()):
                out = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
                print("%-16s %2d  %s" % (base, v, out[:110]))
                bad += v
    print("serialized load pairs: %d" % bad)
    return 0


if __name__ == "__main__":
    sys.exit(main())
