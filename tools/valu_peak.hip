// tools/valu_peak.hip -- what is the integer VALU issue ceiling of an MI355X?
//
// K-SCAN / K-CHIM / K-BC2 are bound by integer vector instructions, so their roofline needs a measured peak, not an
// assumed one (round-1 VERDICT: the guide gives 2 cycles per wave64 instruction per SIMD with >= 2 waves resident, the
// round-1 bench assumed 4).  This program issues streams of ONE instruction -- eight independent register chains per
// wave, no memory traffic inside the loop -- at 1, 2, 4 and 8 waves per SIMD on every CU and reports wave-instructions
// per second, cycles per wave-instruction per SIMD (against the in-kernel clock: s_memtime / s_memrealtime) and the
// chip-wide figure bench.py uses as `valu_issue.peak_ginst_s`.
//
// Build + run (GPU box):  hipcc -O2 --offload-arch=gfx950 tools/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak > profiles/r02/valu_peak.json
//
// Occupancy is pinned with LDS: a block is 256 threads = one wave per SIMD and asks for 160 KiB / W of dynamic LDS, so
// exactly W blocks (W waves per SIMD) are resident per CU; the grid holds 8 rounds of that.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x)                                                                                     \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));        \
            exit(1);                                                                                 \
        }                                                                                            \
    } while (0)

// One kernel per instruction form.  T(k) is the text of ONE instruction on chain register %k (k = 0..7: eight independent
// chains, so a dependent instruction is 8 issue slots away); %8 = b, %9 = c (VGPRs), %10 = sb, %11 = sc (SGPRs).
#define ROWOF(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define REP8(X) X X X X X X X X
struct Entry {
    const char *name;
    void (*fn)(uint32_t *, uint64_t *, int, uint32_t);
    int valu_per_row;   // VALU instructions per 8-instruction row (mixes count only their VALU part)
    int int_peak;       // counts towards the integer peak figure
};
static std::vector<Entry> &registry() {
    static std::vector<Entry> r;
    return r;
}
#define DEF_KERNEL(ID, NAME, ROW, VALU_PER_ROW, INT_PEAK)                                                              \
    __global__ __launch_bounds__(256) void k_##ID(uint32_t *out, uint64_t *stamps, int iters, uint32_t seed) {         \
        extern __shared__ uint32_t lds_pin[]; /* only pins occupancy */                                                \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 9 + 4,             \
                 a5 = a0 * 11 + 5, a6 = a0 * 13 + 6, a7 = a0 * 15 + 7;                                                   \
        uint32_t b = (0x5a5a5a5bu ^ seed) + threadIdx.x, c = 3u + (seed & 1);                                            \
        uint32_t sb = __builtin_amdgcn_readfirstlane(b), sc = __builtin_amdgcn_readfirstlane(c);                         \
        const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                         \
        for (int i = 0; i < iters; ++i)                                                                                \
            asm volatile(REP8(ROW)                                                                                     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)               \
                         : "v"(b), "v"(c), "s"(sb), "s"(sc)                                                            \
                         : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");                        \
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                         \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                      \
        if (threadIdx.x == 0) {                                                                                        \
            stamps[2 * blockIdx.x] = t1 - t0;     /* shader cycles */                                                  \
            stamps[2 * blockIdx.x + 1] = r1 - r0; /* 100 MHz ticks */                                                  \
        }                                                                                                              \
        if (iters < 0) lds_pin[threadIdx.x] = a0;                                                                      \
    }                                                                                                                  \
    static const bool reg_##ID = (registry().push_back({NAME, k_##ID, VALU_PER_ROW, INT_PEAK}), true);

// ---- VOP2, both sources VGPRs
#define T_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n\t"
#define T_OR(k) "v_or_b32 %" #k ", %" #k ", %8\n\t"
#define T_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8\n\t"
#define T_ADD(k) "v_add_u32 %" #k ", %" #k ", %8\n\t"
#define T_SUB(k) "v_sub_u32 %" #k ", %" #k ", %8\n\t"
#define T_MAX(k) "v_max_i32 %" #k ", %" #k ", %8\n\t"
#define T_MIN(k) "v_min_u32 %" #k ", %" #k ", %8\n\t"
#define T_LSHLV(k) "v_lshlrev_b32 %" #k ", %9, %" #k "\n\t"
#define T_LSHRV(k) "v_lshrrev_b32 %" #k ", %9, %" #k "\n\t"
#define T_ASHRV(k) "v_ashrrev_i32 %" #k ", %9, %" #k "\n\t"
#define T_MUL24(k) "v_mul_u32_u24 %" #k ", %" #k ", %8\n\t"
#define T_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n\t"
#define T_ADDCO(k) "v_add_co_u32 %" #k ", vcc, %" #k ", %8\n\t"
DEF_KERNEL(and_vv, "v_and_b32 v,v,v", ROWOF(T_AND), 8, 1)
DEF_KERNEL(or_vv, "v_or_b32 v,v,v", ROWOF(T_OR), 8, 1)
DEF_KERNEL(xor_vv, "v_xor_b32 v,v,v", ROWOF(T_XOR), 8, 1)
DEF_KERNEL(add_vv, "v_add_u32 v,v,v", ROWOF(T_ADD), 8, 1)
DEF_KERNEL(sub_vv, "v_sub_u32 v,v,v", ROWOF(T_SUB), 8, 1)
DEF_KERNEL(max_vv, "v_max_i32 v,v,v", ROWOF(T_MAX), 8, 1)
DEF_KERNEL(min_vv, "v_min_u32 v,v,v", ROWOF(T_MIN), 8, 1)
DEF_KERNEL(lshl_vv, "v_lshlrev_b32 v,v,v", ROWOF(T_LSHLV), 8, 1)
DEF_KERNEL(lshr_vv, "v_lshrrev_b32 v,v,v", ROWOF(T_LSHRV), 8, 1)
DEF_KERNEL(ashr_vv, "v_ashrrev_i32 v,v,v", ROWOF(T_ASHRV), 8, 1)
DEF_KERNEL(mul24_vv, "v_mul_u32_u24 v,v,v", ROWOF(T_MUL24), 8, 1)
DEF_KERNEL(cndmask, "v_cndmask_b32 v,v,v,vcc", ROWOF(T_CNDMASK), 8, 1)
DEF_KERNEL(addco, "v_add_co_u32 v,vcc,v,v", ROWOF(T_ADDCO), 8, 1)
// ---- VOP2 with a scalar register, an inline constant, a literal
#define T_ANDS(k) "v_and_b32 %" #k ", %10, %" #k "\n\t"
#define T_ANDI(k) "v_and_b32 %" #k ", 15, %" #k "\n\t"
#define T_ANDL(k) "v_and_b32 %" #k ", 0x12345678, %" #k "\n\t"
#define T_ADDI(k) "v_add_u32 %" #k ", 1, %" #k "\n\t"
#define T_LSHLI(k) "v_lshlrev_b32 %" #k ", 2, %" #k "\n\t"
#define T_LSHRI(k) "v_lshrrev_b32 %" #k ", 2, %" #k "\n\t"
#define T_LSHLS(k) "v_lshlrev_b32 %" #k ", %11, %" #k "\n\t"
DEF_KERNEL(and_sv, "v_and_b32 v,s,v", ROWOF(T_ANDS), 8, 1)
DEF_KERNEL(and_iv, "v_and_b32 v,15,v (inline)", ROWOF(T_ANDI), 8, 1)
DEF_KERNEL(and_lv, "v_and_b32 v,0x12345678,v (literal)", ROWOF(T_ANDL), 8, 1)
DEF_KERNEL(add_iv, "v_add_u32 v,1,v (inline)", ROWOF(T_ADDI), 8, 1)
DEF_KERNEL(lshl_iv, "v_lshlrev_b32 v,2,v (inline)", ROWOF(T_LSHLI), 8, 1)
DEF_KERNEL(lshr_iv, "v_lshrrev_b32 v,2,v (inline)", ROWOF(T_LSHRI), 8, 1)
DEF_KERNEL(lshl_sv, "v_lshlrev_b32 v,s,v", ROWOF(T_LSHLS), 8, 1)
// ---- VOP1
#define T_MOV(k) "v_mov_b32 %" #k ", %8\n\t"
#define T_MOVS(k) "v_mov_b32 %" #k ", %10\n\t"
#define T_NOT(k) "v_not_b32 %" #k ", %" #k "\n\t"
#define T_BFREV(k) "v_bfrev_b32 %" #k ", %" #k "\n\t"
#define T_FFBH(k) "v_ffbh_u32 %" #k ", %" #k "\n\t"
#define T_FFBL(k) "v_ffbl_b32 %" #k ", %" #k "\n\t"
DEF_KERNEL(mov_v, "v_mov_b32 v,v", ROWOF(T_MOV), 8, 1)
DEF_KERNEL(mov_s, "v_mov_b32 v,s", ROWOF(T_MOVS), 8, 1)
DEF_KERNEL(not_v, "v_not_b32 v,v", ROWOF(T_NOT), 8, 1)
DEF_KERNEL(bfrev, "v_bfrev_b32 v,v", ROWOF(T_BFREV), 8, 1)
DEF_KERNEL(ffbh, "v_ffbh_u32 v,v", ROWOF(T_FFBH), 8, 1)
DEF_KERNEL(ffbl, "v_ffbl_b32 v,v", ROWOF(T_FFBL), 8, 1)
// ---- VOPC (result in VCC) + a consumer-free stream
#define T_CMP(k) "v_cmp_lt_u32 vcc, %" #k ", %8\n\t"
#define T_CMPS(k) "v_cmp_lt_u32 s[40:41], %" #k ", %8\n\t"
DEF_KERNEL(cmp_vcc, "v_cmp_lt_u32 vcc,v,v", ROWOF(T_CMP), 8, 1)
DEF_KERNEL(cmp_sgpr, "v_cmp_lt_u32 s[..],v,v (e64)", ROWOF(T_CMPS), 8, 1)
// ---- VOP3 (three sources or 64-bit encoding only)
#define T_MAX3(k) "v_max3_i32 %" #k ", %" #k ", %8, %9\n\t"
#define T_MIN3(k) "v_min3_u32 %" #k ", %" #k ", %8, %9\n\t"
#define T_MED3(k) "v_med3_i32 %" #k ", %" #k ", %8, %9\n\t"
#define T_MAD24(k) "v_mad_i32_i24 %" #k ", %" #k ", %8, %9\n\t"
#define T_MADU24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n\t"
#define T_ALIGNBIT(k) "v_alignbit_b32 %" #k ", %" #k ", %8, %9\n\t"
#define T_ALIGNBITI(k) "v_alignbit_b32 %" #k ", %" #k ", %8, 2\n\t"
#define T_BFE(k) "v_bfe_u32 %" #k ", %" #k ", %8, %9\n\t"
#define T_BFEI(k) "v_bfe_u32 %" #k ", %" #k ", 3, 5\n\t"
#define T_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n\t"
#define T_OR3(k) "v_or3_b32 %" #k ", %" #k ", %8, %9\n\t"
#define T_BFI(k) "v_bfi_b32 %" #k ", %" #k ", %8, %9\n\t"
#define T_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", %9, %8\n\t"
#define T_LSHLADDI(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %8\n\t"
#define T_LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 2, %8\n\t"
#define T_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n\t"
#define T_XAD(k) "v_xad_u32 %" #k ", %" #k ", %8, %9\n\t"
#define T_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n\t"
#define T_SAD(k) "v_sad_u32 %" #k ", %" #k ", %8, %9\n\t"
#define T_BCNT(k) "v_bcnt_u32_b32 %" #k ", %" #k ", %8\n\t"
#define T_MBCNT(k) "v_mbcnt_lo_u32_b32 %" #k ", %" #k ", %8\n\t"
#define T_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n\t"
#define T_ADD64(k) "v_add_u32_e64 %" #k ", %" #k ", %8\n\t"
#define T_LSHL64(k) "v_lshlrev_b64 v[20:21], %9, v[20:21]\n\t"
#define T_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
#define T_FADD(k) "v_add_f32 %" #k ", %" #k ", %8\n\t"
DEF_KERNEL(max3, "v_max3_i32 v,v,v,v", ROWOF(T_MAX3), 8, 1)
DEF_KERNEL(min3, "v_min3_u32 v,v,v,v", ROWOF(T_MIN3), 8, 1)
DEF_KERNEL(med3, "v_med3_i32 v,v,v,v", ROWOF(T_MED3), 8, 1)
DEF_KERNEL(mad24, "v_mad_i32_i24 v,v,v,v", ROWOF(T_MAD24), 8, 1)
DEF_KERNEL(madu24, "v_mad_u32_u24 v,v,v,v", ROWOF(T_MADU24), 8, 1)
DEF_KERNEL(alignbit, "v_alignbit_b32 v,v,v,v", ROWOF(T_ALIGNBIT), 8, 1)
DEF_KERNEL(alignbit_i, "v_alignbit_b32 v,v,v,2", ROWOF(T_ALIGNBITI), 8, 1)
DEF_KERNEL(bfe, "v_bfe_u32 v,v,v,v", ROWOF(T_BFE), 8, 1)
DEF_KERNEL(bfe_i, "v_bfe_u32 v,v,3,5", ROWOF(T_BFEI), 8, 1)
DEF_KERNEL(and_or, "v_and_or_b32 v,v,v,v", ROWOF(T_ANDOR), 8, 1)
DEF_KERNEL(or3, "v_or3_b32 v,v,v,v", ROWOF(T_OR3), 8, 1)
DEF_KERNEL(bfi, "v_bfi_b32 v,v,v,v", ROWOF(T_BFI), 8, 1)
DEF_KERNEL(lshl_add, "v_lshl_add_u32 v,v,v,v", ROWOF(T_LSHLADD), 8, 1)
DEF_KERNEL(lshl_add_i, "v_lshl_add_u32 v,v,2,v", ROWOF(T_LSHLADDI), 8, 1)
DEF_KERNEL(lshl_or_i, "v_lshl_or_b32 v,v,2,v", ROWOF(T_LSHLOR), 8, 1)
DEF_KERNEL(add3, "v_add3_u32 v,v,v,v", ROWOF(T_ADD3), 8, 1)
DEF_KERNEL(xad, "v_xad_u32 v,v,v,v", ROWOF(T_XAD), 8, 1)
DEF_KERNEL(perm, "v_perm_b32 v,v,v,v", ROWOF(T_PERM), 8, 1)
DEF_KERNEL(sad, "v_sad_u32 v,v,v,v", ROWOF(T_SAD), 8, 1)
DEF_KERNEL(bcnt, "v_bcnt_u32_b32 v,v,v", ROWOF(T_BCNT), 8, 1)
DEF_KERNEL(mbcnt, "v_mbcnt_lo_u32_b32 v,v,v", ROWOF(T_MBCNT), 8, 1)
DEF_KERNEL(mul_lo, "v_mul_lo_u32 v,v,v", ROWOF(T_MULLO), 8, 1)
DEF_KERNEL(add_e64, "v_add_u32_e64 v,v,v (VOP3 encoding of a VOP2 op)", ROWOF(T_ADD64), 8, 1)
DEF_KERNEL(fma, "v_fma_f32 v,v,v,v", ROWOF(T_FMA), 8, 0)
DEF_KERNEL(fadd, "v_add_f32 v,v,v", ROWOF(T_FADD), 8, 0)
// ---- packed 16-bit
#define T_PKADD(k) "v_pk_add_i16 %" #k ", %" #k ", %8\n\t"
#define T_PKMAX(k) "v_pk_max_i16 %" #k ", %" #k ", %8\n\t"
#define T_PKMAD(k) "v_pk_mad_i16 %" #k ", %" #k ", %8, %9\n\t"
DEF_KERNEL(pk_add, "v_pk_add_i16 v,v,v", ROWOF(T_PKADD), 8, 0)
DEF_KERNEL(pk_max, "v_pk_max_i16 v,v,v", ROWOF(T_PKMAX), 8, 0)
DEF_KERNEL(pk_mad, "v_pk_mad_i16 v,v,v,v", ROWOF(T_PKMAD), 8, 0)
// ---- cross-lane
#define T_DPP(k) "v_mov_b32_dpp %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define T_DPPADD(k) "v_add_u32_dpp %" #k ", %" #k ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define T_READLANE(k) "v_readlane_b32 s40, %" #k ", 5\n\t"
#define T_READFIRST(k) "v_readfirstlane_b32 s40, %" #k "\n\t"
DEF_KERNEL(mov_dpp, "v_mov_b32_dpp row_shr:1", ROWOF(T_DPP), 8, 0)
DEF_KERNEL(add_dpp, "v_add_u32_dpp row_shr:1", ROWOF(T_DPPADD), 8, 0)
DEF_KERNEL(readlane, "v_readlane_b32 s,v,5", ROWOF(T_READLANE), 8, 0)
DEF_KERNEL(readfirst, "v_readfirstlane_b32 s,v", ROWOF(T_READFIRST), 8, 0)
// ---- mixes: does a second class fill the gaps of the first?
#define T_MIX_VOP3_VOP2(k) "v_max3_i32 %" #k ", %" #k ", %8, %9\n\tv_and_b32 %" #k ", %" #k ", %8\n\t"
#define T_MIX_SALU(k) "v_and_b32 %" #k ", %" #k ", %8\n\ts_add_u32 s4" #k ", s4" #k ", 1\n\t"
#define T_MIX_SALU3(k) "v_max3_i32 %" #k ", %" #k ", %8, %9\n\ts_add_u32 s4" #k ", s4" #k ", 1\n\t"
DEF_KERNEL(mix32, "v_max3_i32 + v_and_b32 (1:1, rate counts both)", ROWOF(T_MIX_VOP3_VOP2), 16, 0)
DEF_KERNEL(mix_salu, "v_and_b32 + independent s_add_u32 (1:1, rate counts VALU only)", ROWOF(T_MIX_SALU), 8, 0)
DEF_KERNEL(mix_salu3, "v_max3_i32 + independent s_add_u32 (1:1, rate counts VALU only)", ROWOF(T_MIX_SALU3), 8, 0)

// ---- how the two classes combine (patterns inside one wave's stream)
#define T_AND_LSHR(k) "v_and_b32 %" #k ", %" #k ", %8\n\tv_lshrrev_b32 %" #k ", 1, %" #k "\n\t"
#define ROW_BLOCKS ROWOF(T_AND) ROWOF(T_MAX3)
#define T_AAAB(k) "v_and_b32 %" #k ", %" #k ", %8\n\tv_xor_b32 %" #k ", %" #k ", %9\n\tv_add_u32 %" #k ", %" #k ", %8\n\tv_max3_i32 %" #k ", %" #k ", %8, %9\n\t"
#define ROW_DEP "v_and_b32 %0, %0, %8\n\tv_xor_b32 %0, %0, %9\n\tv_add_u32 %0, %0, %8\n\tv_or_b32 %0, %0, %9\n\tv_sub_u32 %0, %0, %8\n\tv_lshrrev_b32 %0, 1, %0\n\tv_add_u32 %0, %0, %9\n\tv_xor_b32 %0, %0, %8\n\t"
#define ROW_DEP2 "v_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_xor_b32 %0, %0, %9\n\tv_xor_b32 %1, %1, %9\n\tv_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_or_b32 %0, %0, %9\n\tv_or_b32 %1, %1, %9\n\t"
#define ROW_DEP_MAX3 "v_max3_i32 %0, %0, %8, %9\n\tv_mad_i32_i24 %0, %0, %8, %9\n\tv_max3_i32 %0, %0, %8, %9\n\tv_bfe_u32 %0, %0, 3, 5\n\tv_max3_i32 %0, %0, %8, %9\n\tv_alignbit_b32 %0, %0, %8, 2\n\tv_and_or_b32 %0, %0, %8, %9\n\tv_max3_i32 %0, %0, %8, %9\n\t"
DEF_KERNEL(mix_and_lshr, "v_and_b32 + v_lshrrev_b32 (both 2-cycle forms, 1:1)", ROWOF(T_AND_LSHR), 16, 0)
DEF_KERNEL(mix_blocks, "8 x v_and_b32 then 8 x v_max3_i32 (blocks, rate counts both)", ROW_BLOCKS, 16, 0)
DEF_KERNEL(mix_aaab, "and, xor, add, max3 per chain (3:1, rate counts all four)", ROWOF(T_AAAB), 32, 0)
DEF_KERNEL(dep1, "ONE dependent chain of 2-cycle forms", ROW_DEP, 8, 0)
DEF_KERNEL(dep2, "TWO dependent chains of 2-cycle forms, interleaved", ROW_DEP2, 8, 0)
DEF_KERNEL(dep1_vop3, "ONE dependent chain of 4-cycle forms", ROW_DEP_MAX3, 8, 0)

// ---- round 5: what does a select cost?  v_cndmask_b32 alone reads 18.6 cycles above; is that the instruction, its VCC operand, or the stream?
#define T_CND64(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[40:41]\n\t"
#define ROW_CND1_MAX7 "v_cndmask_b32 %0, %0, %8, vcc\n\t" T_MAX3(1) T_MAX3(2) T_MAX3(3) T_MAX3(4) T_MAX3(5) T_MAX3(6) T_MAX3(7)
#define ROW_CND1_AND7 "v_cndmask_b32 %0, %0, %8, vcc\n\t" T_AND(1) T_AND(2) T_AND(3) T_AND(4) T_AND(5) T_AND(6) T_AND(7)
#define ROW_CND2_MAX6 "v_cndmask_b32 %0, %0, %8, vcc\n\t" T_MAX3(1) T_MAX3(2) T_MAX3(3) "v_cndmask_b32 %4, %4, %8, vcc\n\t" T_MAX3(5) T_MAX3(6) T_MAX3(7)
#define ROW_CMP_CND "v_cmp_lt_u32 vcc, %0, %8\n\t" T_MAX3(1) T_MAX3(2) "v_cndmask_b32 %0, %0, %9, vcc\n\t" T_MAX3(3) T_MAX3(4) T_MAX3(5) T_MAX3(6)
#define ROW_CMP64_CND64 "v_cmp_lt_u32 s[40:41], %0, %8\n\t" T_MAX3(1) T_MAX3(2) "v_cndmask_b32_e64 %0, %0, %9, s[40:41]\n\t" T_MAX3(3) T_MAX3(4) T_MAX3(5) T_MAX3(6)
#define ROW_MAX8 T_MAX3(0) T_MAX3(1) T_MAX3(2) T_MAX3(3) T_MAX3(4) T_MAX3(5) T_MAX3(6) T_MAX3(7)
DEF_KERNEL(cnd_e64, "r5: v_cndmask_b32_e64 v,v,v,s[40:41]", ROWOF(T_CND64), 8, 0)
DEF_KERNEL(cnd1_max7, "r5: 1 v_cndmask (vcc) + 7 v_max3 per row (rate counts all 8)", ROW_CND1_MAX7, 8, 0)
DEF_KERNEL(cnd1_and7, "r5: 1 v_cndmask (vcc) + 7 v_and per row (rate counts all 8)", ROW_CND1_AND7, 8, 0)
DEF_KERNEL(cnd2_max6, "r5: 2 v_cndmask (vcc) + 6 v_max3 per row (rate counts all 8)", ROW_CND2_MAX6, 8, 0)
DEF_KERNEL(cmp_cnd, "r5: v_cmp vcc .. v_cndmask vcc + 6 v_max3 per row (rate counts all 8)", ROW_CMP_CND, 8, 0)
DEF_KERNEL(cmp64_cnd64, "r5: v_cmp s[] .. v_cndmask_e64 s[] + 6 v_max3 per row (rate counts all 8)", ROW_CMP64_CND64, 8, 0)
DEF_KERNEL(max8, "r5: 8 v_max3 per row (reference for the rows above)", ROW_MAX8, 8, 0)

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 8000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const int waves_per_simd[4] = {1, 2, 4, 8};
    const int rounds = 8;
    uint32_t *d_out;
    uint64_t *d_stamps;
    const size_t max_blocks = (size_t)n_cu * 8 * rounds;
    CHECK(hipMalloc(&d_out, max_blocks * 256 * 4));
    CHECK(hipMalloc(&d_stamps, max_blocks * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"compute_units\": %d, \"simds\": %d, \"iters\": %d, \"rows\": [\n", prop.name,
           prop.gcnArchName, n_cu, n_cu * 4, iters);
    double best_int = 0.0, best_vop3 = 0.0;
    std::string best_name;
    bool first = true;
    const char *only = argc > 2 ? argv[2] : nullptr;  // run the entries whose name holds this text (e.g. "r5:")
    for (const Entry &en : registry()) {
        if (only && !strstr(en.name, only)) continue;
        for (int wi = 0; wi < 4; ++wi) {
            const int W = waves_per_simd[wi];
            const size_t lds = (160 * 1024 / W) & ~(size_t)1023;  // W blocks fill a CU's LDS, a (W+1)-th does not fit
            CHECK(hipFuncSetAttribute((const void *)en.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const unsigned grid = (unsigned)(n_cu * W * rounds);
            hipLaunchKernelGGL(en.fn, dim3(grid), dim3(256), lds, 0, d_out, d_stamps, iters / 10, 1u);  // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(en.fn, dim3(grid), dim3(256), lds, 0, d_out, d_stamps, iters, 2u);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint64_t> st(2 * (size_t)grid);
            CHECK(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> clk(grid), cyc(grid);
            for (unsigned g = 0; g < grid; ++g) {
                clk[g] = st[2 * g + 1] ? (double)st[2 * g] / (double)st[2 * g + 1] * 0.1 : 0.0;  // GHz
                cyc[g] = (double)st[2 * g];
            }
            std::sort(clk.begin(), clk.end());
            std::sort(cyc.begin(), cyc.end());
            const double ghz = clk[grid / 2];
            const double valu_per_wave = (double)iters * 8.0 * en.valu_per_row;  // 8 rows per iteration
            const double total = valu_per_wave * 4.0 * grid;
            const double ginst_s = total / (ms * 1e-3) / 1e9;
            // a block's wave shares its SIMD with W-1 others: cycles the SIMD spends per wave-instruction
            const double cyc_per_inst_simd = cyc[grid / 2] / (valu_per_wave * W);
            printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"ginst_per_s\": %.1f, \"clock_ghz\": %.3f, "
                   "\"cycles_per_wave_inst_per_simd\": %.3f}",
                   first ? "" : ",\n", en.name, W, ms, ginst_s, ghz, cyc_per_inst_simd);
            first = false;
            if (en.int_peak && ginst_s > best_int) {
                best_int = ginst_s;
                best_name = std::string(en.name) + " @" + std::to_string(W) + " waves/SIMD";
            }
            if (std::string(en.name) == "v_max3_i32 v,v,v,v" && ginst_s > best_vop3) best_vop3 = ginst_s;
        }
    }
    printf("\n], \"peak_int_ginst_s\": %.1f, \"peak_at\": \"%s\", \"peak_vop3_ginst_s\": %.1f, \"note\": \"wave-instructions per second "
           "over the whole chip; two classes exist: 2-cycle forms and 4-cycle forms (see rows)\"}\n",
           best_int, best_name.c_str(), best_vop3);
    return 0;
}
