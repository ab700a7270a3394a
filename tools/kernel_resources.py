"""Register / scratch / LDS use of every kernel instantiation in fiber_amd/csrc (runs on the build host: hipcc cross-compiles, no GPU needed).

    python tools/kernel_resources.py [tag] > profiles/<tag>_kernel_resources.txt

Compiles each .hip with -Rpass-analysis=kernel-resource-usage (device side only, objects discarded) and prints one row per kernel, kernels
with scratch first.  Names are demangled with c++filt (whose table lacks the bf16 type: it is passed as `half`; the argument list is dropped anyway)."""
import concurrent.futures as cf
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fiber_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FILT = "/usr/bin/c++filt"


def one(path):
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-c", path, "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=SRC)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1), "file": os.path.basename(path)[:-4]}
            rows.append(cur)
            continue
        for key, pat in (("sgpr", r"SGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return rows


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    files = sorted(glob.glob(os.path.join(SRC, "*.hip")))
    with cf.ThreadPoolExecutor(8) as ex:
        rows = [r for rs in ex.map(one, files) for r in rs]
    names = subprocess.run([FILT], input="\n".join(r["name"].replace("DF16b", "Dh") for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["name"] = re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
    rows.sort(key=lambda r: (-r.get("scratch", 0), r["file"], r["name"]))
    print(f"# Register / scratch use of every instantiated kernel (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, ROCm 7.2), {tag}.")
    print("# scratch = bytes per lane of spill space; kernels with scratch > 0 are listed first.  occ = waves per SIMD the registers admit.\n")
    print(f"{'file':10s} {'scratch':>7s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'LDS':>7s} {'occ':>3s}  kernel")
    for r in rows:
        print(f"{r['file']:10s} {r.get('scratch', 0):7d} {r.get('vgpr', 0):5d} {r.get('agpr', 0):5d} {r.get('sgpr', 0):5d} {r.get('lds', 0):7d} {r.get('occ', 0):3d}  {r['name']}")
    n_scr = sum(1 for r in rows if r.get("scratch", 0))
    print(f"\n# {len(rows)} kernels, {n_scr} with scratch")


if __name__ == "__main__":
    main()
