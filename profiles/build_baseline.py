"""Build the library of an EARLIER commit as a variant (brdf_nerf_amd/build/<tag>/libbrdfnerf_hip.so) so that profiles/ab_kernels.py
can time it against the working tree in one process (cdna_hip_programming.md rule 24).

    python profiles/build_baseline.py <commit> <tag>

The old kernel sources are compiled with the flags THEIR build.py used; error.cpp and the public header come from the working
tree (the loader checks every symbol the current header declares)."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    commit, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "brdf_nerf_amd", "build", tag)
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(f"git -C {ROOT} archive {commit} brdf_nerf_amd/csrc brdf_nerf_amd/build.py include | tar -x -C {td}", shell=True)
        shutil.copy(os.path.join(ROOT, "brdf_nerf_amd", "csrc", "error.cpp"), os.path.join(td, "brdf_nerf_amd", "csrc", "error.cpp"))
        if not os.path.exists(os.path.join(td, "brdf_nerf_amd", "csrc", "diag.h")):     # (round 5: error.cpp takes bn_build_flags from diag.h)
            shutil.copy(os.path.join(ROOT, "brdf_nerf_amd", "csrc", "diag.h"), os.path.join(td, "brdf_nerf_amd", "csrc", "diag.h"))
        old_hdr = open(os.path.join(td, "include", "brdfnerf_hip.h")).read()
        if "bn_build_flags" not in old_hdr:     # declare what the new error.cpp defines, keep the old structs
            old_hdr = old_hdr.replace("const char *bn_last_error(void);", "const char *bn_last_error(void);\nconst char *bn_build_flags(void);")
            open(os.path.join(td, "include", "brdfnerf_hip.h"), "w").write(old_hdr)
        sys.path.insert(0, os.path.join(td, "brdf_nerf_amd"))
        import importlib.util
        spec = importlib.util.spec_from_file_location("old_build", os.path.join(td, "brdf_nerf_amd", "build.py"))
        ob = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ob)
        lib = ob.build(force=True, out=os.path.join(out, "libbrdfnerf_hip.so"), tag=tag)
        # the old build.py writes its objects under ITS tree (the temp dir): only the .so is kept
        print(lib)


if __name__ == "__main__":
    main()
