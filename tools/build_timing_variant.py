#!/usr/bin/env python3
"""Builds build/exp/lib_TIMING.so: the kernel library with wall-clock phase stamps in conv_mfma_kernel
(experiment only; tools/conv_timeline.py reads them).  The product library is untouched."""
import os, subprocess, glob
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(R, "dahitra_amd/csrc/conv_mfma_impl.h")).read()
def rep(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new, 1)
rep("template <typename T, int KS, int STRIDE, int NT, int RW, int DIL, bool PF, bool FAST>\n__global__",
    '''__device__ long long g_ts[8192 * 16];
__device__ long long g_ld[8192 * 16];
#define TS(k) do { if (tid == 0 && blockIdx.x + blockIdx.y * gridDim.x < 8192) g_ts[(blockIdx.x + blockIdx.y * gridDim.x) * 16 + (k)] = (long long)wall_clock64(); } while (0)
template <typename T, int KS, int STRIDE, int NT, int RW, int DIL, bool PF, bool FAST>
__global__''')
rep("""                    rh[i] = *reinterpret_cast<const uint4*>(
                        xin + ((size_t)(iy * p.W + ix) * p.Cin + c0) * sizeof(T) + q * 16);
            }
        }""", """                    rh[i] = *reinterpret_cast<const uint4*>(
                        xin + ((size_t)(iy * p.W + ix) * p.Cin + c0) * sizeof(T) + q * 16);
            }
            if (c0 == 0 && tid == 0 && blockIdx.x + blockIdx.y * gridDim.x < 8192) g_ld[(blockIdx.x + blockIdx.y * gridDim.x) * 16 + i] = (long long)wall_clock64();
        }""")
rep("""                rw[i] = *reinterpret_cast<const uint4*>(
                    wgt + ((size_t)(tap * p.CoutPad + co0 + co) * p.Cin + c0) * sizeof(T) + q * 16);
            }""", """                rw[i] = *reinterpret_cast<const uint4*>(
                    wgt + ((size_t)(tap * p.CoutPad + co0 + co) * p.Cin + c0) * sizeof(T) + q * 16);
            }
            if (c0 == 0 && tid == 0 && blockIdx.x + blockIdx.y * gridDim.x < 8192) g_ld[(blockIdx.x + blockIdx.y * gridDim.x) * 16 + NHV + i] = (long long)wall_clock64();""")
rep("    if (PF) fetch(0);\n    for (int c0 = 0; c0 < p.Cin; c0 += CK) {\n        if (!PF) fetch(c0);\n        commit();\n        __syncthreads();\n",
    '''    TS(0);
    if (PF) fetch(0);
    for (int c0 = 0; c0 < p.Cin; c0 += CK) {
        if (!PF) fetch(c0);
        if (c0 == 0) TS(1);
        commit();
        if (c0 == 0) TS(2);
        __syncthreads();
        if (c0 == 0) TS(3);
        if (c0 == CK) TS(5);
''')
rep("        __syncthreads();\n    }\n\n    // ---- epilogue ----", "        if (c0 == 0) TS(4);\n        __syncthreads();\n    }\n    TS(6);\n\n    // ---- epilogue ----")
rep("                    st4(trow + s * 16, v);\n                }\n", "                    st4(trow + s * 16, v);\n                }\n                if (r < 3) TS(13 + r);\n")
rep("    if (p.stats) {\n        // reduce over the 16 pixel lanes", "    TS(7);\n    if (p.stats) {\n        // reduce over the 16 pixel lanes")
rep("    if (p.stats || wide) __syncthreads();\n    if (wide) {", "    TS(8);\n    if (p.stats || wide) __syncthreads();\n    TS(9);\n    if (wide) {")
rep("    if (p.stats) {\n        const float* red = reinterpret_cast<const float*>(smem);", "    TS(10);\n    if (p.stats) {\n        const float* red = reinterpret_cast<const float*>(smem);")
rep('''                p.stats[((size_t)which * p.CoutPad + co0 + c) * gridDim.x + blockIdx.x] = t;   // [2][CoutPad][tiles]
        }
    }
}
''', '''                p.stats[((size_t)which * p.CoutPad + co0 + c) * gridDim.x + blockIdx.x] = t;   // [2][CoutPad][tiles]
        }
    }
    TS(11);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TS(12);
}
extern "C" int dh_debug_ts(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ts), (size_t)n * 8); }
extern "C" int dh_debug_ld(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ld), (size_t)n * 8); }
extern "C" int dh_debug_clear() { static long long z[8192 * 16]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ts), z, sizeof(z)); }
''')
# the stamped kernel is the header; the read-back entry points go into the bf16 translation unit (its copy of g_ts)
dbg = "\n".join(l for l in s.split("\n") if l.startswith('extern "C" int dh_debug_'))
s = "\n".join(l for l in s.split("\n") if not l.startswith('extern "C" int dh_debug_'))
E = os.path.join(R, "build/exp")
os.makedirs(E, exist_ok=True)
open(os.path.join(E, "conv_mfma_impl.h"), "w").write(s)
import shutil
shutil.copy(os.path.join(R, "dahitra_amd/csrc/conv_mfma_bf16.hip"), os.path.join(E, "conv_mfma_bf16.hip"))
open(os.path.join(E, "conv_mfma_bf16.hip"), "a").write("\n" + dbg)
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(R, "include"), "-I" + os.path.join(R, "dahitra_amd/csrc"), "-Wno-unused-result"]
o = os.path.join(E, "conv_mfma_bf16_T.o")
subprocess.check_call(["hipcc"] + flags + ["-c", os.path.join(E, "conv_mfma_bf16.hip"), "-o", o])
others = [x for x in glob.glob(os.path.join(R, "build/obj/*.o")) if os.path.basename(x) != "conv_mfma_bf16.o"]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(R, "build/exp/lib_TIMING.so"), o] + others)
print("built build/exp/lib_TIMING.so")
