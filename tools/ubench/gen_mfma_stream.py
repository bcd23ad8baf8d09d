"""Generates + builds a micro-benchmark of the bf16 core's k-step shape (one wave per SIMD, 4 waves per workgroup):
  per k-step: 1 ds_read_b128 (A fragment, read 4 k-steps ahead) + 2 x v_mfma_f32_32x32x16_bf16 (two accumulators)
  + F filler instructions of a chosen kind.  Prints shader cycles per MFMA for each variant.
Operands are bf16-looking random data (zeros would flatter the power draw); next to the cycles the tool prints the
shader clock the governor settled on (shader cycles / 100 MHz real-time ticks) and the wall time of the loop.
Usage (GPU box): python tools/ubench/gen_mfma_stream.py
"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))

KSTEPS = 32          # per asm block (2 "tiles" of 16)
def body(fill_kind, per_kstep, acc="a", bsrc="v", dsread=True, first_gap=None, rotate_b=False, bias_c=False, glds=False, barrier=False,
         real_order=False, nop=True, ds_first=False, tile_extras=False, x_doubles=True, x_bias=True, x_salu=True):
    L = []
    acc0, acc1 = ("a[0:15]", "a[16:31]") if acc == "a" else ("v[128:143]", "v[144:159]")
    b0, b1 = ("v[16:19]", "v[20:23]") if bsrc == "v" else ("a[64:67]", "a[68:71]")
    fills = []
    if fill_kind == "valu":
        fills = ["v_add_f32 v%d, v%d, v%d" % (100 + (i % 8), 110 + (i % 8), 120 + (i % 8)) for i in range(per_kstep)]
    elif fill_kind == "accread":
        fills = ["v_accvgpr_read_b32 v%d, a%d" % (100 + (i % 8), 32 + (i % 16)) for i in range(per_kstep)]
    elif fill_kind == "epi":     # realistic quarter(s): 2 accread, 2 add, cvt_pk, pk_max per 6
        for qd in range(per_kstep // 6):
            r = 100 + 2 * (qd % 4)
            fills += ["v_accvgpr_read_b32 v%d, a%d" % (r, 32 + 2 * (qd % 8)), "v_accvgpr_read_b32 v%d, a%d" % (r + 1, 33 + 2 * (qd % 8)),
                      "v_add_f32 v%d, v%d, v110" % (r, r), "v_add_f32 v%d, v%d, v111" % (r + 1, r + 1),
                      "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (r, r, r + 1), "v_pk_max_i16 v%d, v%d, 0" % (112 + qd % 4, r)]
    elif fill_kind == "epi4":    # the shipped quarter: 2 accread | cvt_pk, pk_max
        r = 100
        fills = ["v_accvgpr_read_b32 v%d, a%d" % (r, 32), "v_accvgpr_read_b32 v%d, a%d" % (r + 1, 33),
                 "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (r + 2, r, r + 1), "v_pk_max_i16 v%d, v%d, 0" % (112, r + 2)]
    elif fill_kind == "epi_v":   # same work but the accumulators being read live in VGPRs (no accvgpr_read)
        for qd in range(per_kstep // 4):
            r = 100 + 2 * (qd % 4)
            fills += ["v_add_f32 v%d, v%d, v110" % (r, 160 + 2 * (qd % 8)), "v_add_f32 v%d, v%d, v111" % (r + 1, 161 + 2 * (qd % 8)),
                      "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (r, r, r + 1), "v_pk_max_i16 v%d, v%d, 0" % (112 + qd % 4, r)]
    half = (len(fills) + 1) // 2 if first_gap is None else first_gap
    for s in range(KSTEPS):
        q = 4 * (s % 4)
        if rotate_b:     # a different B operand every k-step, as the real layer loop (activations of 16 k-steps x 2 groups)
            b0 = "v[%d:%d]" % (24 + 8 * (s % 8), 27 + 8 * (s % 8))
            b1 = "v[%d:%d]" % (28 + 8 * (s % 8), 31 + 8 * (s % 8))
        c0, c1 = acc0, acc1
        if bias_c and s % 16 == 0:   # a tile's first MFMAs take the bias as C from another register set
            c0 = c1 = "a[72:87]"
        if dsread:
            L.append("s_waitcnt lgkmcnt(3)")
        if real_order and nop:
            L.append("s_nop 0")
        L.append("v_mfma_f32_32x32x16_bf16 %s, v[%d:%d], %s, %s" % (acc0, q, q + 3, b0, c0))
        L += fills[:half]
        if glds and s % 4 == 1:
            L += ["s_mov_b32 m0, %2", "s_nop 0", "global_load_lds_dwordx4 %3, %1 offset:0"]
        L.append("v_mfma_f32_32x32x16_bf16 %s, v[%d:%d], %s, %s" % (acc1, q, q + 3, b1, c1))
        if real_order and not ds_first:
            L += fills[half:]
            if dsread:
                L.append("ds_read_b128 v[%d:%d], %%0 offset:%d" % (q, q + 3, 1024 * (s % 16)))
        else:
            if dsread:
                L.append("ds_read_b128 v[%d:%d], %%0 offset:%d" % (q, q + 3, 1024 * (s % 16)))
            L += fills[half:]
        if tile_extras:   # what a 16-k-step tile of the real layer loop carries besides the steady-state k-step
            sl = s % 16
            if x_doubles and sl in (4, 8, 12):      # second quarter of the double k-steps
                L += fills
            if x_bias and sl in (12, 13):        # next tile's bias, two 16-byte LDS reads per k-step straight into AGPRs
                L += ["ds_read_b128 a[72:75], %0 offset:16384", "ds_read_b128 a[76:79], %0 offset:16400"]
            if x_salu == "spread":    # the same bookkeeping cut into pieces over k-steps 14, 15, 0, 2, 3
                sal = lambda n, b: ["s_add_u32 s%d, s%d, %d" % (40 + (b + i) % 7, 40 + (b + i) % 7, i + 1) for i in range(n)]
                if sl == 14: L += sal(4, 0) + ["v_mov_b32 v124, v126", "v_mov_b32 v126, v127"]
                if sl == 15: L += sal(6, 1)
                if sl == 0: L += sal(6, 2)
                if sl == 2: L += sal(3, 3)
                if sl == 3: L += sal(2, 4) + ["v_add_u32 v127, s41, v125"]
            if x_salu is True and sl == 14:              # advance(): ring-slot bookkeeping
                L += ["s_add_u32 s40, s40, 1", "s_cmp_eq_u32 s40, 6", "s_cselect_b32 s40, 0, s40", "s_lshl_b32 s41, s40, 14", "s_add_u32 s41, s41, 0x5800",
                      "v_add_u32 v124, s41, v125", "v_add_u32 v126, s41, v125"]
            if x_salu is True and sl == 15:              # cursor_update(): ~10 scalar instructions
                L += ["s_add_u32 s42, s42, 1", "s_cmp_eq_u32 s42, 6", "s_cselect_b32 s42, 0, s42", "s_cmp_eq_u32 s43, 1", "s_cselect_b32 s44, 1, 0",
                      "s_sub_u32 s43, s43, 1", "s_add_u32 s45, s45, 0x4000", "s_addc_u32 s46, s46, 0", "s_cmp_lg_u32 s44, 0", "s_cselect_b32 s43, 76, s43"]
        if barrier and s % 16 == 14:
            L += ["s_waitcnt vmcnt(8)", "s_barrier"]
    return "\\n\\t".join(L)

VARIANTS = [
    ("mfma only (acc AGPR)", dict(fill_kind=None, per_kstep=0, dsread=False)),
    ("mfma only (acc VGPR)", dict(fill_kind=None, per_kstep=0, dsread=False, acc="v")),
    ("+ ds_read", dict(fill_kind=None, per_kstep=0)),
    ("+ ds_read, B from AGPR", dict(fill_kind=None, per_kstep=0, bsrc="a")),
    ("+ ds_read + 6 valu/kstep", dict(fill_kind="valu", per_kstep=6)),
    ("+ ds_read + 8 valu/kstep", dict(fill_kind="valu", per_kstep=8)),
    ("+ ds_read + 10 valu/kstep", dict(fill_kind="valu", per_kstep=10)),
    ("+ ds_read + 12 valu/kstep", dict(fill_kind="valu", per_kstep=12)),
    ("+ ds_read + 6 accread/kstep", dict(fill_kind="accread", per_kstep=6)),
    ("+ ds_read + 1 quarter (6)/kstep", dict(fill_kind="epi", per_kstep=6)),
    ("+ ds_read + 2 quarters (12)/kstep", dict(fill_kind="epi", per_kstep=12)),
    ("+ ds_read + 1 quarter, acc VGPR (4)/kstep", dict(fill_kind="epi_v", per_kstep=4, acc="v")),
    ("+ ds_read + 2 quarters, acc VGPR (8)/kstep", dict(fill_kind="epi_v", per_kstep=8, acc="v")),
    ("+ ds_read + 1 quarter all in 2nd gap", dict(fill_kind="epi", per_kstep=6, first_gap=0)),
    ("real k-step order, quarter of 4 (2+2)", dict(fill_kind="epi4", per_kstep=4, real_order=True)),
    ("real order without the s_nop", dict(fill_kind="epi4", per_kstep=4, real_order=True, nop=False)),
    ("real order, ds_read first in gap 2", dict(fill_kind="epi4", per_kstep=4, real_order=True, ds_first=True)),
    ("real order, no nop, ds_read first", dict(fill_kind="epi4", per_kstep=4, real_order=True, nop=False, ds_first=True)),
    ("  + B operand rotates over 64 regs", dict(fill_kind="epi4", per_kstep=4, real_order=True, rotate_b=True)),
    ("  + bias as C of a tile's first MFMAs", dict(fill_kind="epi4", per_kstep=4, real_order=True, rotate_b=True, bias_c=True)),
    ("  + LDS-DMA piece every 4th k-step", dict(fill_kind="epi4", per_kstep=4, real_order=True, rotate_b=True, bias_c=True, glds=True)),
    ("  + vmcnt wait + s_barrier every 16th", dict(fill_kind="epi4", per_kstep=4, real_order=True, rotate_b=True, bias_c=True, glds=True, barrier=True)),
    ("shipped order (ds_read first) + DMA + barrier", dict(fill_kind="epi4", per_kstep=4, real_order=True, ds_first=True, rotate_b=True, bias_c=True, glds=True, barrier=True)),
    ("  + per-tile extras (doubles, bias reads, SALU)", dict(fill_kind="epi4", per_kstep=4, real_order=True, ds_first=True, rotate_b=True, bias_c=True, glds=True, barrier=True, tile_extras=True)),
    ("    extras: only the double quarters", dict(fill_kind="epi4", per_kstep=4, real_order=True, ds_first=True, rotate_b=True, bias_c=True, glds=True, barrier=True, tile_extras=True, x_bias=False, x_salu=False)),
    ("    extras: only the bias reads", dict(fill_kind="epi4", per_kstep=4, real_order=True, ds_first=True, rotate_b=True, bias_c=True, glds=True, barrier=True, tile_extras=True, x_doubles=False, x_salu=False)),
    ("    extras: SALU bookkeeping spread over 5 k-steps", dict(fill_kind="epi4", per_kstep=4, real_order=True, ds_first=True, rotate_b=True, bias_c=True, glds=True, barrier=True, tile_extras=True, x_doubles=False, x_bias=False, x_salu="spread")),
    ("    all extras, SALU spread", dict(fill_kind="epi4", per_kstep=4, real_order=True, ds_first=True, rotate_b=True, bias_c=True, glds=True, barrier=True, tile_extras=True, x_salu="spread")),
    ("    extras: only the SALU bookkeeping", dict(fill_kind="epi4", per_kstep=4, real_order=True, ds_first=True, rotate_b=True, bias_c=True, glds=True, barrier=True, tile_extras=True, x_doubles=False, x_bias=False)),
]

clob = '"s40", "s41", "s42", "s43", "s44", "s45", "s46", "scc", ' + ", ".join('"v%d"' % i for i in range(0, 96)) + ", " + ", ".join('"v%d"' % i for i in range(100, 196)) + ", " + ", ".join('"a%d"' % i for i in range(0, 88))
INIT = "\\n\\t".join("v_xor_b32 v%d, 0x%x, %%0" % (r, (r * 0x01230123) & 0x007f007f) for r in list(range(0, 96)))
src = ['#include <hip/hip_runtime.h>', '#include <stdio.h>', 'typedef __attribute__((address_space(3))) char lds_char;']
for k, (name, kw) in enumerate(VARIANTS):
    src.append('''__global__ __launch_bounds__(256, 1) void k%d(unsigned long long* out, int iters, const char* gsrc0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned addr = (unsigned)(size_t)(lds_char*)smem + (threadIdx.x & 63) * 16;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* gsrc = gsrc0 + wv * 4096;
  unsigned ldsdst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)smem) + 32768 + wv * 4096;
  unsigned voff = (threadIdx.x & 63) * 16;
  // bf16-looking operands (sign, exponent near 1.0, random mantissa): the matrix pipe's power draw -- and with it the
  // clock the governor settles on -- depends on the data toggling, zeros would flatter it
  for (int i = threadIdx.x; i < 16384; i += 256) { unsigned x = (i * 2654435761u) ^ (i >> 3); ((unsigned*)smem)[i] = (x & 0x807f807fu) | 0x3f003f00u | ((x >> 9) & 0x00800080u); }
  unsigned pat = ((threadIdx.x * 40503u) & 0x807f807fu) | 0x3f003f00u;
  asm volatile("%s" ::"v"(pat) : %s);
  __syncthreads();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    asm volatile("%s" ::"v"(addr), "s"(gsrc), "s"(ldsdst), "v"(voff) : %s, "memory");
  }
  asm volatile("s_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15" ::: "memory");
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}''' % (k, INIT, clob, body(**kw), clob))
src.append('int main() { unsigned long long* d; hipMalloc(&d, 16); char* g; hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20); unsigned long long h[2]; const int iters = 4000;')
for k, (name, kw) in enumerate(VARIANTS):
    src.append('  hipFuncSetAttribute((const void*)k%d, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);' % k)
    src.append('  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k%d, dim3(256), dim3(256), 65536, 0, d, iters, g); hipDeviceSynchronize(); }' % k)
    src.append('  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("%%-46s %%6.2f cycles/MFMA   %%.3f GHz  %%.0f us\\n", "%s", (double)h[0] / (iters * %d.0), (double)h[0] / ((double)h[1] * 10.0), (double)h[1] / 100.0);' % (name, 2 * KSTEPS))
src.append('  return 0; }')
path = os.path.join(HERE, "mfma_stream_gen.hip")
open(path, "w").write("\n".join(src))
exe = os.path.join(HERE, "mfma_stream")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-w", "-o", exe, path])
if "--build-only" not in sys.argv:
    subprocess.check_call([exe])
