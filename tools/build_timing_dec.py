#!/usr/bin/env python3
"""Builds build/exp/lib_TIMINGD.so: wall-clock phase sums per workgroup inside dec_bwd_kernel (experiment only)."""
import os, subprocess, glob
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(R, "dahitra_amd/csrc/decoder_fused.hip")).read()
def rep(old, new, cnt=1):
    global s
    assert s.count(old) == cnt, (s.count(old), old)
    s = s.replace(old, new)
rep("template <int MLP>\n__global__ __launch_bounds__(256) void dec_bwd_kernel(DecArgs p) {",
    "__device__ long long g_td[4096 * 8];\n#define NOW() ((long long)wall_clock64())\ntemplate <int MLP>\n__global__ __launch_bounds__(256) void dec_bwd_kernel(DecArgs p) {\n    const long long t_start = NOW();")
rep("    stage(sW1T, MLP + 8, p.w1T, D, MLP, tid);\n    __syncthreads();\n\n    const f32x4 zero4", "    stage(sW1T, MLP + 8, p.w1T, D, MLP, tid);\n    __syncthreads();\n    const long long t_staged = NOW();\n    long long t_chain = 0, t_prod = 0;\n\n    const f32x4 zero4")
rep("    for (int it = 0; it < p.rows_per_block / 128; ++it) {\n        // packed operands", "    for (int it = 0; it < p.rows_per_block / 128; ++it) {\n        const long long t_it = NOW();\n        // packed operands")
rep("        // ---- pixel-reduction products over this wave's 32 pixels (K = 32), through wave-private LDS tiles ----", "        const long long t_mid = NOW(); t_chain += t_mid - t_it;\n        // ---- pixel-reduction products over this wave's 32 pixels (K = 32), through wave-private LDS tiles ----")
rep("        __syncthreads();\n    }\n\n    // ---- combine the 4 wavefronts deterministically, then", "        __syncthreads();\n        t_prod += NOW() - t_mid;\n    }\n    const long long t_loop = NOW();\n\n    // ---- combine the 4 wavefronts deterministically, then")
rep("        out[i] = ((red[i] + red[P::SIZE + i]) + red[2 * P::SIZE + i]) + red[3 * P::SIZE + i];\n}", "        out[i] = ((red[i] + red[P::SIZE + i]) + red[2 * P::SIZE + i]) + red[3 * P::SIZE + i];\n    if (tid == 0 && blockIdx.x < 4096) { long long* o = g_td + blockIdx.x * 8; o[0] = t_start; o[1] = t_staged; o[2] = t_chain; o[3] = t_prod; o[4] = t_loop; o[5] = NOW(); o[6] = p.rows_per_block / 128; }\n}\nextern \"C\" int dh_debug_td(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_td), (size_t)n * 8); }")
E = os.path.join(R, "build/exp"); os.makedirs(E, exist_ok=True)
src = os.path.join(E, "dec_timing.hip"); open(src, "w").write(s)
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(R, "include"), "-I" + os.path.join(R, "dahitra_amd/csrc"), "-Wno-unused-result"]
o = os.path.join(E, "dec_T.o")
subprocess.check_call(["hipcc"] + flags + ["-c", src, "-o", o])
others = [x for x in glob.glob(os.path.join(R, "build/obj/*.o")) if os.path.basename(x) != "decoder_fused.o"]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(E, "lib_TIMINGD.so"), o] + others)
print("built build/exp/lib_TIMINGD.so")
