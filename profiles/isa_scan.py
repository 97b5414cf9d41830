"""Vector instructions inside the MFMA loops of a kernel, from hipcc's assembly (no GPU needed).

    python profiles/isa_scan.py field_fwd.hip 'field_fwd_kernelIDF16bLi4ELi2ELi8ELb1ELb0'  [-DFLAG ...]

Compiles csrc/<file> to gfx950 assembly with the file's product flags (brdf_nerf_amd/build.py FILE_FLAGS) and, for every kernel
whose mangled name matches the pattern, lists the loops that contain MFMAs: MFMA count, instruction count, scratch accesses and
the vector (v_*) instructions besides the MFMAs.  In these kernels a vector instruction beside the MFMAs is not free (round 4,
profiles/r04_ablation.txt items 16-18, 20): the tool is how the bias sums of wgrad256 (~36 per 8 MFMAs), the per-read LDS
address adds of the chain GEMM and the accumulator copies of its tail steps were found.  Loop detection is crude (a backward
branch to a label): nested loops and spin-waits show up as overlapping ranges."""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from brdf_nerf_amd import build as B  # noqa: E402


def main():
    src, pat = sys.argv[1], sys.argv[2]
    defs = [a for a in sys.argv[3:] if a.startswith("-D")]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [B.HIPCC] + B.FLAGS + list(B.FILE_FLAGS.get(src, ())) + defs + ["-x", "hip", "--cuda-device-only", "-S",
                                                                            os.path.join(B.CSRC, src), "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    for k in re.split(r"\n(?=_Z[\w]+:\s*; @)", text):
        m = re.match(r"(_Z[\w]+):", k)
        if not m or not re.search(pat, m.group(1)):
            continue
        lines = k.split("\n")
        labels = {mm.group(1): i for i, l in enumerate(lines) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
        print("==", m.group(1))
        for i, l in enumerate(lines):
            mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                a = labels[mm.group(1)]
                body = [x.split()[0] for x in lines[a:i] if x.strip() and not x.strip().startswith((";", "."))]
                nm = sum("v_mfma" in x for x in body)
                if nm >= 8 and len(body) < 600:
                    c = Counter(x for x in body if x.startswith("v_") and "mfma" not in x)
                    print(f"  lines {a}-{i}: {nm} mfma, {len(body)} instructions, {sum('scratch' in x for x in body)} scratch, "
                          f"{sum(c.values())} vector: {dict(c.most_common(8))}")


if __name__ == "__main__":
    main()
