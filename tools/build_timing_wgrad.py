#!/usr/bin/env python3
"""Builds build/exp/lib_TIMINGW.so: phase sums per workgroup inside conv_wgrad_kernel (experiment only)."""
import os, subprocess, glob
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(R, "dahitra_amd/csrc/conv_wgrad.hip")).read()
def rep(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new, 1)
rep("template <typename T, int KS, int STRIDE, int IT, bool TR, int DIL, int CTT, int CIG, bool DYT = false>\n__global__",
    '''__device__ long long g_tw[8192 * 16];
#define NOW() ((long long)wall_clock64())
template <typename T, int KS, int STRIDE, int IT, bool TR, int DIL, int CTT, int CIG, bool DYT = false>
__global__''')
rep("    if (kz < ntiles) fetch();\n    for (int tile = kz; tile < ntiles; tile += p.splitk) {\n        commit();\n        __syncthreads();\n        if (tile + p.splitk < ntiles) fetch();\n",
    '''    long long tsum[6] = {0, 0, 0, 0, 0, 0};
    const long long tstart = NOW();
    if (kz < ntiles) fetch();
    tsum[0] = NOW() - tstart;
    for (int tile = kz; tile < ntiles; tile += p.splitk) {
        long long t0 = NOW();
        commit();
        long long t1 = NOW(); tsum[1] += t1 - t0;
        __syncthreads();
        t0 = NOW(); tsum[2] += t0 - t1;
        if (tile + p.splitk < ntiles) fetch();
        t1 = NOW(); tsum[3] += t1 - t0;
''')
rep("        __syncthreads();\n    }\n\n    if constexpr (KSPLIT > 1) {",
    '''        t0 = NOW(); tsum[4] += t0 - t1;
        __syncthreads();
        tsum[5] += NOW() - t0;
    }
    const long long tloop = NOW();

    if constexpr (KSPLIT > 1) {''')
rep('''                    if (okij[i][j]) out[t * tstride + j * p.Cin + i * istep] = acc[t][i][j];
    }
}
''', '''                    if (okij[i][j]) out[t * tstride + j * p.Cin + i * istep] = acc[t][i][j];
    }
    {
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (tid == 0 && lin < 8192) {
            long long* o = g_tw + lin * 16;
            for (int k = 0; k < 6; ++k) o[k] = tsum[k];
            o[6] = tstart; o[7] = tloop; o[8] = NOW(); o[9] = (ntiles - kz + p.splitk - 1) / p.splitk;
        }
    }
}
extern "C" int dh_debug_tw(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tw), (size_t)n * 8); }
extern "C" int dh_debug_tw_clear() { static long long z[8192 * 16]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tw), z, sizeof(z)); }
''')
os.makedirs(os.path.join(R, "build/exp"), exist_ok=True)
src = os.path.join(R, "build/exp/wgrad_timing.hip")
open(src, "w").write(s)
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(R, "include"), "-I" + os.path.join(R, "dahitra_amd/csrc"), "-Wno-unused-result"]
subprocess.check_call(["hipcc"] + flags + ["-c", src, "-o", os.path.join(R, "build/exp/wgrad_TIMING.o")])
objs = [o for o in glob.glob(os.path.join(R, "build/obj/*.o")) if not o.endswith("conv_wgrad.o")]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(R, "build/exp/lib_TIMINGW.so"), os.path.join(R, "build/exp/wgrad_TIMING.o")] + objs)
print("built build/exp/lib_TIMINGW.so")
