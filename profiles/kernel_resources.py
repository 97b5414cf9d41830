"""Register / scratch / occupancy table of every kernel in the library (hipcc -Rpass-analysis=kernel-resource-usage).

    python profiles/kernel_resources.py [-DFLAG ...] [--filter substr]

Compiles each csrc/*.hip to an object in a scratch directory (nothing in-tree is touched) and prints one line per kernel:
VGPRs, AGPRs, scratch bytes per lane, occupancy (waves per SIMD), LDS.  A kernel that spills shows scratch > 0.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from brdf_nerf_amd import build as B  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def main():
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    filt = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--filter=")]
    rows = []
    with tempfile.TemporaryDirectory() as td:
        procs = []
        for src in B.sources():
            if not src.endswith(".hip"):
                continue
            cmd = [B.HIPCC] + B.FLAGS + list(B.FILE_FLAGS.get(os.path.basename(src), ())) + ["-D" + d for d in defs] + ["-x", "hip", "-c", src, "-o", os.path.join(td, os.path.basename(src) + ".o"),
                                                                   "-Rpass-analysis=kernel-resource-usage"]
            procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        for p in procs:
            log, _ = p.communicate()
            cur = None
            for line in log.splitlines():
                m = re.search(r"remark: (?:Function Name: (\S+)|\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+))", line)
                if not m:
                    continue
                if m.group(1):
                    cur = {"name": m.group(1)}
                    rows.append(cur)
                elif cur is not None:
                    cur[m.group(2).strip()] = int(m.group(3))
    names = demangle([r["name"] for r in rows])
    print(f"{'VGPR':>5} {'AGPR':>5} {'scratch':>8} {'occ':>4} {'LDS':>7}  kernel")
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", ""))
        if filt and not any(f in n for f in filt):
            continue
        print(f"{r.get('VGPRs', 0):>5} {r.get('AGPRs', 0):>5} {r.get('ScratchSize', 0):>8} {r.get('Occupancy', 0):>4} {r.get('LDS Size', 0):>7}  {n}")


if __name__ == "__main__":
    main()
