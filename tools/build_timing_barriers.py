#!/usr/bin/env python3
"""Builds build/exp/lib_TIMINGB.so: a wall-clock stamp after every __syncthreads() of ONE kernel and, optionally, of
device functions it calls (experiment only).
usage: build_timing_barriers.py <file.hip> '<text that starts the kernel definition>' ['<text that starts a function>' ...]
Read back with dh_debug_tb (tools/barrier_timeline.py)."""
import os, re, subprocess, glob, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fname, starts = sys.argv[1], sys.argv[2:]
s = open(os.path.join(R, "dahitra_amd/csrc", fname)).read()
STAMP = "__syncthreads(); DH_TS();"
total = 0
for n, start in enumerate(starts):
    i0 = s.index(start)
    b0 = s.index("{", i0)
    i1 = s.index("\n}\n", i0) + 3
    body = s[b0:i1]
    total += body.count("__syncthreads();")
    body = body.replace("__syncthreads();", STAMP)
    if n == 0:      # the kernel: reset the counter, first stamp, final stamp
        body = "{\n    if (threadIdx.x == 0) { g_tc[blockIdx.x + gridDim.x * blockIdx.y & 255] = 0; } DH_TS();" + body[1:]
        j = body.rindex("}")
        body = body[:j] + "    DH_TS();\n}\n"
    s = s[:b0] + body + s[i1:]
decl = """
__device__ long long g_tb[256 * 64];
__device__ int g_tc[256];
#define DH_TS() do { if (threadIdx.x == 0) { const int l_ = (blockIdx.x + gridDim.x * blockIdx.y) & 255; const int k_ = g_tc[l_]; if (k_ < 64) { g_tb[l_ * 64 + k_] = (long long)wall_clock64(); g_tc[l_] = k_ + 1; } } } while (0)
"""
# declarations right after the last #include
inc = [m.end() for m in re.finditer(r"#include [^\n]*\n", s)][-1]
s = s[:inc] + decl + s[inc:] + """
extern "C" int dh_debug_tb(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tb), (size_t)n * 8); }
extern "C" int dh_debug_tb_clear() { static long long z[256 * 64]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tb), z, sizeof(z)); }
"""
E = os.path.join(R, "build/exp"); os.makedirs(E, exist_ok=True)
src = os.path.join(E, "barrier_timing.hip"); open(src, "w").write(s)
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(R, "include"), "-I" + os.path.join(R, "dahitra_amd/csrc"), "-Wno-unused-result"]
if fname.split(".")[0] in ("decoder_fused", "encoder_fused", "tokens"):
    flags += ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
o = os.path.join(E, "barrier_T.o")
subprocess.check_call(["hipcc"] + flags + ["-c", src, "-o", o])
others = [x for x in glob.glob(os.path.join(R, "build/obj/*.o")) if os.path.basename(x) != fname.replace(".hip", ".o")]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(E, "lib_TIMINGB.so"), o] + others)
print("built build/exp/lib_TIMINGB.so with", total, "stamped barriers")
