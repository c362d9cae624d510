"""Dev tool: for kernels whose mangled name contains argv[1], report scratch ops inside the MFMA region of the ISA."""
import subprocess, sys, re
pat = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "/root/repo/duodiff_amd/csrc/gemm.hip"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", src,
                "-o", "/tmp/isa_check.s"], check=True, stderr=subprocess.DEVNULL)
lines = open("/tmp/isa_check.s").read().split("\n")
for i, l in enumerate(lines):
    if l.endswith(":") or re.match(r"^_Z\S+:\s", l):
        name = l.split(":")[0]
        if pat in name and not name.startswith("."):
            end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
            body = lines[i:end]
            mf = [j for j, b in enumerate(body) if "v_mfma" in b]
            if not mf:
                continue
            sc = [j for j in range(mf[0], mf[-1]) if "scratch_" in body[j]]
            print(f"{name[:90]}: {len(body)} lines, {len(mf)} mfma, scratch ops inside mfma region: {len(sc)}, "
                  f"total scratch ops {sum('scratch_' in b for b in body)}")
