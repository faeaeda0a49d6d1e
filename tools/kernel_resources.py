"""Register / LDS / scratch use and occupancy of every kernel of the library, from the compiler's own remarks
(-Rpass-analysis=kernel-resource-usage).  usage: python tools/kernel_resources.py [extra hipcc flags...]"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "bibim_renderer_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
       "-fPIC", "-I" + os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(src, "bibim_hip.hip"),
       "-o", "/dev/null"] + sys.argv[1:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark:\s+([A-Za-z][\w \[\]/]*?): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip().split("(bbr::")[0].split("(HIP")[0]
    except OSError:
        return n
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'waves/SIMD':>10s}")
for r in rows:
    n = demangle(r["name"]).replace("void bbr::", "")
    print(f"{n[:58]:58s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('LDS Size [bytes/block]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>10s}")
