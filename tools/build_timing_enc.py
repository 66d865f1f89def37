#!/usr/bin/env python3
"""Builds build/exp/lib_TIMINGE.so: a wall-clock stamp after every barrier of encoder_bwd_kernel (experiment only)."""
import os, re, subprocess, glob
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(R, "dahitra_amd/csrc/encoder_fused.hip")).read()
i0 = s.index("__global__ __launch_bounds__(256) void encoder_bwd_kernel(EncArgs a) {")
i1 = s.index("\n}\n", i0) + 3      # end of the kernel body
body = s[i0:i1]
k = [0]
def stamp(m):
    k[0] += 1
    return "__syncthreads(); if (threadIdx.x == 0 && k_ts < 63) ts[k_ts++] = (long long)wall_clock64(); /*S%d*/" % k[0]
body = body.replace("void encoder_bwd_kernel(EncArgs a) {", "void encoder_bwd_kernel(EncArgs a) {\n    long long ts[64]; int k_ts = 0; ts[k_ts++] = (long long)wall_clock64();", 1)
body = re.sub(r"__syncthreads\(\);", stamp, body)
# write-out at the end of the kernel: last closing brace of the kernel
j = body.rindex("}")
body = body[:j] + "    if (threadIdx.x == 0 && blockIdx.x < 64) { for (int q = 0; q < 64; ++q) g_te[blockIdx.x * 64 + q] = q < k_ts ? ts[q] : 0; }\n}\n"
s = s[:i0] + "__device__ long long g_te[64 * 64];\n" + body + 'extern "C" int dh_debug_te(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_te), (size_t)n * 8); }\n' + s[i1:]
E = os.path.join(R, "build/exp"); os.makedirs(E, exist_ok=True)
src = os.path.join(E, "enc_timing.hip"); open(src, "w").write(s)
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(R, "include"), "-I" + os.path.join(R, "dahitra_amd/csrc"), "-Wno-unused-result", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
o = os.path.join(E, "enc_T.o")
subprocess.check_call(["hipcc"] + flags + ["-c", src, "-o", o])
others = [x for x in glob.glob(os.path.join(R, "build/obj/*.o")) if os.path.basename(x) != "encoder_fused.o"]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(E, "lib_TIMINGE.so"), o] + others)
print("built build/exp/lib_TIMINGE.so with", k[0], "stamped barriers")
