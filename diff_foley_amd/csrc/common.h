// Common device helpers for the gfx950 (MI355X / CDNA4) kernels of the Diff-Foley sampling path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- MFMA operand type.  The library is compiled twice from the same sources: with bf16 operands (default,
// libdfengine.so -- BASELINE config "bf16 UNet") and with -DDF_OPERAND_F16 (libdfengine_f16.so: fp16 operands, 3 more
// mantissa bits at the same MFMA rate; out-of-range values saturate at +-65504).  Accumulation, the residual stream,
// norm statistics and softmax are fp32 in both.  Names keep the historical "bf" prefix: bf16_t = raw operand bits,
// f2bf/bf2f/pack_bf2 = float <-> operand conversions (round-to-nearest-even, native v_cvt_pk_{bf16,f16}_f32).
typedef uint16_t bf16_t;  // raw operand bits (bf16 or fp16)
#if defined(DF_OPERAND_F16)
typedef _Float16 op_scalar;
#define DF_OPERAND_NAME "f16"
#define DF_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define DF_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
__device__ __forceinline__ float op_clamp(float f) { return __builtin_fminf(__builtin_fmaxf(f, -65504.f), 65504.f); }
#else
typedef __bf16 op_scalar;
#define DF_OPERAND_NAME "bf16"
#define DF_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define DF_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
__device__ __forceinline__ float op_clamp(float f) { return f; }
#endif
typedef __attribute__((ext_vector_type(8))) op_scalar bf16x8;
typedef __attribute__((ext_vector_type(2))) op_scalar op_x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define DF_WAVE 64

// ---- Write-through stores of activations (round 4).  The 8 XCDs' L2s are not coherent with each other, so the release at the end
// of every kernel writes back whatever the kernel left dirty in L2 -- AFTER the last block has finished: a tail proportional to the
// output bytes (tools/gemm_bench.py, DF_GEMM_DBG=2: the 10.5 MB of a full-resolution fp32 output cost 2.7 of the 8.1 us of the
// 8192 x 320 x 320 projection and 3.6 of the 22.4 us of the 320 -> 320 conv).  `sc1` (agent scope) sends the line on towards
// memory when it is stored, under the blocks that are still computing, and the kernel-end write-back finds nothing to do:
// 8.1 -> 6.8 us and 22.4 -> 20.9 us on those two.  Nothing is lost for the reader: the next kernel's acquire invalidates L2 anyway.
// (`nt` gives half of that.)  -DDF_NO_WT_STORES compiles the plain stores back in (A/B).
// Hazard: gfx950 needs 2 wait states between a VMEM store of more than 64 bits and a VALU write of its data VGPRs, and the
// hazard recogniser does not see into inline asm -- the dwordx4 forms carry their own `s_nop 1` (= 2 wait states); the <= 64-bit
// forms need none.  tools/check_store_hazard.py scans the built libraries for a dwordx4 sc1 store without it.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DF_NO_WT_STORES)
__device__ __forceinline__ void st_wt(float4* p, const float4& v) {
  const f32x4 t = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void st_wt(uint4* p, const uint4& v) {
  const u32x4 t = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void st_wt(uint2* p, const uint2& v) {
  const u32x2 t = {v.x, v.y};
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void st_wt(float2* p, const float2& v) {
  const f32x2 t = {v.x, v.y};
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void st_wt(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_wt(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
#else
template <class T> __device__ __forceinline__ void st_wt(T* p, const T& v) { *p = v; }
#endif

// ---- Packed-FP32 operand select on src1: broken on this hardware (round 6).  `v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32` whose LOW
// result lane takes the HIGH register of src1 (`op_sel:[x,1]`) read that operand as ZERO in lanes 48-63 while another wavefront of
// the same SIMD executes MFMAs: d + 0, d * 0, a * 0 + d.  Reproduced outside the engine by experiments/pk_opsel_probe.hip
// (profiles/r6_pk_opsel_probe.txt: 2 or 4 blocks per CU with an MFMA loop next to the instruction: 10^4-10^5 wrong low results per
// 2 * 10^9, all in the last lane quarter; none with one block per CU, none without MFMAs in the neighbour, none for the select on
// src0 or src2).  The compiler emits the src1 form whenever a packed op broadcasts the odd element of a register pair from its
// second operand; tools/check_pk_opsel.py (run by __graft_entry__.build()) fails the build on any such encoding in the libraries.
// Where the source needs that broadcast it says so in the src0 form:   (lo, hi) += bhi   with (blo, bhi) a register pair.
__device__ __forceinline__ void pk_add_hi(float& lo, float& hi, float blo, float bhi) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 d = {lo, hi};
  const f32x2 b = {blo, bhi};
  asm("v_pk_add_f32 %0, %1, %0 op_sel:[1,0] op_sel_hi:[1,1]" : "+v"(d) : "v"(b));
  lo = d.x;
  hi = d.y;
#else
  lo += bhi;
  hi += bhi;
#endif
}

// ---- Kernel-entry touch (round 5; rebuilt in round 6): kernel-argument lines and the kernel's own code through the VECTOR memory path.
// A kernel's arguments live in the kernarg segment -- new memory at every launch, so the scalar cache and L2 miss on every 64-B
// line of it -- and the compiler s_loads a field where it is first used: the prologue of a kernel with a large argument block
// (GemmParams: 6 lines) is a CHAIN of scalar round trips, one per new line (six `s_load ...; s_waitcnt lgkmcnt(0)` groups in front
// of the first operand request of a GEMM block).  Scalar loads cannot be issued ahead without being waited for (they return out
// of order: every use waits for all of them -- fetching the whole struct in one batch measured +0.5 us per op IN THE PLAN although
// it is 1 k cycles faster warm, experiments/kernarg_batched_fetch_and_prologue_split.patch.txt).  One vector load, lane i reading a
// dword of line i, brings all lines into L2 side by side, and the later scalar loads of the chain become L2 hits: +1.8 % end to end
// for the GEMM kernels alone (same box, 270.7 -> 275.7 steps/s).
// The same for the kernel's own CODE: the next DF_CODE_TOUCH x 4 KB of instructions behind the program counter (the instruction
// cache is cold at every kernel boundary and fetches line by line; +0.4 %).  Never past the code: `_etext` is the linker's own symbol
// for the end of THIS code object's .text section (lld defines it when it is referenced; tools/check_code_touch.py verifies, for every
// built code object, that it exists and equals .text's end) -- lanes whose line would reach it stay off.  (Rounds 4-5 bounded the touch by
// a .bss variable of the code object, i.e. relied on the loader mapping everything between .text and .bss: round-5 advisor finding.)
//
// ROUND 6 -- the touch loads are ORDINARY loads the compiler can see.  Rounds 4-5 issued them from inline asm into one "+v" register
// that "stays reserved until the final wait".  It does not: the compiler does not know a load is in flight to that register, so under
// register pressure it copies the (to it, settled) value to an AGPR or another VGPR and re-uses the register -- for an operand-request
// offset, an MFMA fragment, an accumulator -- and a touch load that returns later (a cold code line is an HBM miss, ~1-2 us) overwrites
// live data.  In the shipped builds the copies sat > 1200 instructions behind the touch in eight GEMM instantiations (rare: a loaded
// box, a second process); with GELU's erf on v_rcp_f32 (-DDF_ERF_RCP: shorter scalar-fallback code in every EPI_ANY kernel) the
// allocator moved the copy of gemm_bf16_kernel<128,128,2,2,4,0,EPI_ANY> -- the cost-model plan's st.ffproj -- to 220 instructions
// behind the touch and re-used the register for request offsets: every forward differed from the last and stored garbage / NaN a few
// ops later (tools/race_hunt.py: deterministic with either half of the touch compiled out, or with the touch out of the MODE 0
// kernels only; round 5 had blamed the GEGLU epilogue).  tools/check_touch_regs.py now scans the built code objects for any
// hand-issued load whose destination is written or copied later; the touch itself no longer needs it: with a compiler-visible load
// SIInsertWaitcnts puts the covering s_waitcnt in front of ANY instruction that reads or overwrites the register.
#if !defined(DF_CODE_TOUCH)
#define DF_CODE_TOUCH 4
#endif
struct DfTouch { int v[DF_CODE_TOUCH > 0 ? DF_CODE_TOUCH : 1]; };
typedef const int __attribute__((address_space(1)))* df_gptr_t;      // global address space: global_load, not flat_load
#if defined(__HIP_DEVICE_COMPILE__)
extern "C" __device__ const char _etext[];
__device__ __forceinline__ DfTouch df_entry_touch(int kernarg_bytes) {
  DfTouch t;
#pragma unroll
  for (int k = 0; k < (DF_CODE_TOUCH > 0 ? DF_CODE_TOUCH : 1); ++k) t.v[k] = 0;
#if !defined(DF_NO_KERNARG_TOUCH)
  const int lane = (int)threadIdx.x;
  const int nka = (kernarg_bytes + 63) / 64;        // kernel-argument lines: lanes [0, nka) of the first load
  const char* const kseg = reinterpret_cast<const char*>(__builtin_amdgcn_kernarg_segment_ptr());
#if DF_CODE_TOUCH > 0
  unsigned long pc;
  asm volatile("s_getpc_b64 %0" : "=s"(pc));
  const long room = (long)(reinterpret_cast<unsigned long>(&_etext[0]) - pc);      // bytes of this code object's .text behind the pc
  const int avail = room > (long)(DF_CODE_TOUCH * 4096) ? DF_CODE_TOUCH * 4096 : (int)room;
#pragma unroll
  for (int k = 0; k < DF_CODE_TOUCH; ++k) {
    // load k: lane i reads a dword of code line k * 64 + i; in load 0 the first nka lanes read the kernel-argument lines instead
    const int off = (k * 64 + lane) * 64;
    const bool is_ka = (k == 0) && lane < nka;
#if defined(DF_NO_KERNARG_LINES)
    const bool ka_on = false;
#else
    const bool ka_on = true;
#endif
    // (a GLOBAL-segment load with the default cache policy: a generic pointer would make it a FLAT load -- counted in lgkmcnt too,
    // so the first scalar-load wait of the prologue would wait for the code line -- and `nt` would mark the lines evict-first in L2)
    const unsigned long a = (is_ka && ka_on) ? reinterpret_cast<unsigned long>(kseg) + (unsigned long)(lane * 64) : pc + (unsigned long)off;
    const bool on = lane < 64 && ((is_ka && ka_on) || (!is_ka && off + 64 <= avail));
    if (on) t.v[k] = *reinterpret_cast<df_gptr_t>(a);
  }
#else
  if (lane < nka) t.v[0] = *reinterpret_cast<df_gptr_t>(reinterpret_cast<unsigned long>(kseg) + (unsigned long)(lane * 64));
#endif
  asm volatile("" ::: "memory");      // the loads stay in front of everything below (nothing may sink them to their use)
#endif
  return t;
}
// The one use of the loaded dwords: an empty asm that takes them as operands -- the compiler's own wait sits in front of it.
__device__ __forceinline__ void df_entry_touch_end(const DfTouch& t) {
#pragma unroll
  for (int k = 0; k < (DF_CODE_TOUCH > 0 ? DF_CODE_TOUCH : 1); ++k) asm volatile("" ::"v"(t.v[k]) : "memory");
}
#else     // host pass: the kernels' bodies are parsed, never run
__device__ __forceinline__ DfTouch df_entry_touch(int) { return DfTouch(); }
__device__ __forceinline__ void df_entry_touch_end(const DfTouch&) {}
#endif

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const f32x2 f = {op_clamp(lo), op_clamp(hi)};
  const op_x2 v = __builtin_convertvector(f, op_x2);
  return *reinterpret_cast<const uint32_t*>(&v);
}
// same without the fp16 saturation, for values known to be bounded (softmax probabilities in the attention loops)
__device__ __forceinline__ uint32_t pack_bf2_bounded(float lo, float hi) {
  const f32x2 f = {lo, hi};
  const op_x2 v = __builtin_convertvector(f, op_x2);
  return *reinterpret_cast<const uint32_t*>(&v);
}
// fp16 build: conversions that clamp in HARDWARE.  MODE.FP16_OVFL (bit 23 of the wavefront's MODE register) makes an fp32 -> fp16
// conversion whose finite result overflows return +-65504 instead of +-inf (checked on gfx950: 1e6 -> 0x7bff, 70000 -> 0x7bff, inf
// stays inf) -- what op_clamp does with a v_max + v_med3 per value.  A kernel that calls df_fp16_hw_clamp() first may pack with
// pack_bf2_hw; everything else keeps pack_bf2.  (df_debug_saturations counts stored +-65504 either way.)
__device__ __forceinline__ void df_fp16_hw_clamp() {
#if defined(DF_OPERAND_F16) && defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1 /* hwreg(HW_REG_MODE, 23, 1) */, 1);
#endif
}
__device__ __forceinline__ uint32_t pack_bf2_hw(float lo, float hi) { return pack_bf2_bounded(lo, hi); }

__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)(pack_bf2(f, 0.f) & 0xFFFFu); }
__device__ __forceinline__ float bf2f(uint16_t h) {
#if defined(DF_OPERAND_F16)
  return (float)*reinterpret_cast<const _Float16*>(&h);
#else
  return __uint_as_float(((uint32_t)h) << 16);
#endif
}
// x * sigmoid(x) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (v_div_scale / v_div_fmas / v_div_fixup
// + Newton steps, ~12 instructions): the result is rounded to the operand type right after, and the GroupNorm that applies it is a
// latency chain whose tail this sits on (round 5).
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32 round-off level): 1 rcp + 1 exp + 6 FMA instead of
// the ~50-instruction libm erff -- the GEGLU epilogue evaluates it for every FF hidden unit.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  // The reciprocal is the hardware's v_rcp_f32 (1 ulp), not __frcp_rn: the correctly rounded form compiles to a whole IEEE division
  // per hidden unit (v_div_scale x 2, v_rcp, four FMAs, v_div_fmas, v_div_fixup: 128 of the 584 VALU instructions of the persistent
  // GEGLU kernel's epilogue block; with v_rcp_f32 the compiler also packs the block: 261), + 0.55 % end to end (round 5, same box).
  // Round 5 had to revert it: the full bf16 model then stored garbage / NaN a few ops behind a generic GEGLU launch, at a different
  // op from run to run.  Round 6 found the cause, and it was not this arithmetic: the kernel-entry touch loads of that era landed in
  // a register the compiler had re-used (see df_entry_touch above); the shorter code only moved the register allocation of
  // st.ffproj's kernel.  -DDF_ERF_DIV compiles the division back in (A/B).
#if defined(DF_ERF_DIV)
  const float t = __frcp_rn(1.0f + 0.3275911f * ax);
#else
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
#endif
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  const float y = 1.0f - poly * t * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf(float x) {  // exact-erf GELU (F.gelu default)
  return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
}

// The value of the partner lane 32 away (lane ^ 32) without the LDS pipe: v_permlane32_swap_b32 (gfx950) exchanges the upper half of its
// first operand with the lower half of its second, so after swapping two copies of v every lane holds its own value in one register and
// the partner's in the other.  (__shfl_xor(v, 32) is a ds_bpermute round trip and an lgkmcnt(0) on the dependent chain; the builtin
// __builtin_amdgcn_permlane32_swap folded a following max of its two results away in this toolchain, hence the asm.  s_nop 1 on both
// sides: the swap reads and writes VGPRs of neighbouring VALU instructions outside the compiler's hazard tracking.)
__device__ __forceinline__ void half_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float half_max(float v) {      // max(v, v of lane ^ 32)
  float a = v, b = v;
  half_swap(a, b);
  return fmaxf(a, b);
}

// Lane exchanges inside a group of 2 / 4 / 8 neighbouring lanes as DPP operands of the consuming VALU instruction (no ds_bpermute round
// trip on the dependent chain): quad_perm [1,0,3,2] = lane ^ 1, quad_perm [2,3,0,1] = lane ^ 2, row_half_mirror = lane 7 - i of the same
// 8 lanes (the other quad: after the two quad steps every lane of a quad holds the quad's total, so this completes 8 lanes).
__device__ __forceinline__ float dpp_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_half_mirror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Same sum with the four in-row steps as DPP modifiers on VALU adds (no LDS-pipe round trips); lanes of a 16-lane
// row all hold the row total, then two cross-row exchanges finish the wave.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));  // row_ror:8
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `red` = 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
// Wave total without the LDS pipe: the four in-row DPP steps, then row_bcast:15 (row r's lane 15 into row r + 1, rows 1 and 3) and
// row_bcast:31 (lane 31 into rows 2 and 3) leave the total in lane 63; v_readlane hands it to every lane through an SGPR.  (The
// __shfl_xor(16 / 32) steps of wave_sum_dpp are ds_bpermute round trips, ~100 cycles each on a latency chain.)
__device__ __forceinline__ float wave_sum_dpp_bcast(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1, 3
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// block_sum_dpp on a `red` array that nobody has read yet (first use in the kernel, or a second array): no leading barrier.
// `red` = 16 floats, 16-byte aligned.  The slots of wavefronts that do not exist are zeroed by wavefront 0 in front of the same barrier
// and all 16 are read as four ds_read_b128 in flight together, summed in slot order (x + 0.0f == x: the same bits as the loop over
// the live slots).  (Round 5: the loop over blockDim.x / 64 slots compiled to one 8-slot pass + up to seven DEPENDENT single-slot
// LDS round trips, ~700 cycles per reduction at 14 wavefronts, two reductions per norm.)
__device__ __forceinline__ float block_sum_dpp_fresh(float v, float* red) {
  v = wave_sum_dpp_bcast(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  else if (threadIdx.x >= nw && threadIdx.x < 16) red[threadIdx.x] = 0.f;
  __syncthreads();
  const float4* r4 = reinterpret_cast<const float4*>(red);
  const float4 a = r4[0], b = r4[1], c = r4[2], d = r4[3];
  float t = 0.f;
  t += a.x; t += a.y; t += a.z; t += a.w;
  t += b.x; t += b.y; t += b.z; t += b.w;
  t += c.x; t += c.y; t += c.z; t += c.w;
  t += d.x; t += d.y; t += d.z; t += d.w;
  return t;
}
// The same with the in-wave steps as DPP adds (wave_sum_dpp): 2 LDS-pipe exchanges per wavefront instead of 6.
__device__ __forceinline__ float block_sum_dpp(float v, float* red) {
  v = wave_sum_dpp(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
