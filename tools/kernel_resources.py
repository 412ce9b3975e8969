"""Per-kernel register / LDS / scratch / occupancy table of the extension (hipcc -Rpass-analysis=kernel-resource-usage).

usage: python tools/kernel_resources.py [extra hipcc flags]   (CPU only: hipcc cross-compiles gfx950)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from layerskip_amd.build import HIPCC_FLAGS  # noqa: E402

CSRC = os.path.join(ROOT, "layerskip_amd", "csrc")


def main():
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    with tempfile.TemporaryDirectory() as td:
        rows = []
        for src in srcs:
            cmd = ["/opt/rocm/bin/hipcc"] + HIPCC_FLAGS + ["-fPIC", "-shared", "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(td, "x.so"),
                                              src] + sys.argv[1:]
            txt = subprocess.run(cmd, capture_output=True, text=True).stderr
            for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
                name = b.split("\n")[0].strip().split()[0]
                dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()

                def g(key):
                    m = re.search(re.escape(key) + r": (\S+)", b)
                    return m.group(1) if m else "?"
                rows.append((dn[:100], g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"),
                             g("LDS Size [bytes/block]")))
        print(f"{'kernel':100s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scr':>4s} {'occ':>4s} {'lds':>7s}")
        for r in rows:
            print(f"{r[0]:100s} {r[1]:>5s} {r[2]:>5s} {r[3]:>5s} {r[4]:>4s} {r[5]:>4s} {r[6]:>7s}")


if __name__ == "__main__":
    main()
