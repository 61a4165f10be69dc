// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the integer VALU ops the kernels lean on.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x
#define DEFK(name, ASM)                                                                             \
__global__ void __launch_bounds__(256) name(uint32_t* out, int iters, uint32_t seed)                 \
{                                                                                                   \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u + 1u, a2 = a0 * 5u + 2u, a3 = a0 * 7u + 3u;       \
    uint32_t b = seed * 0x9E3779B9u + threadIdx.x, c = b ^ 0x55AA55AAu;                              \
    for (int i = 0; i < iters; i++) {                                                               \
        REP16(asm volatile(ASM(0) "\n" ASM(1) "\n" ASM(2) "\n" ASM(3)                                 \
              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");)                           \
    }                                                                                               \
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345678u) out[0] = a0;                                             \
}
// operand numbering: %0..%3 accumulators, %4 = b, %5 = c
#define A_XOR(n)     "v_xor_b32 %" #n ", %4, %" #n
#define A_BCNT(n)    "v_bcnt_u32_b32 %" #n ", %4, %" #n
#define A_ADD(n)     "v_add_u32 %" #n ", %4, %" #n
#define A_MUL24(n)   "v_mul_u32_u24 %" #n ", %4, %" #n
#define A_MAD24(n)   "v_mad_u32_u24 %" #n ", %4, %5, %" #n
#define A_MULLO(n)   "v_mul_lo_u32 %" #n ", %4, %" #n
#define A_PKADD(n)   "v_pk_add_u16 %" #n ", %4, %" #n
#define A_PKMIN(n)   "v_pk_min_u16 %" #n ", %4, %" #n
#define A_PKMAD(n)   "v_pk_mad_u16 %" #n ", %4, %5, %" #n
#define A_PKMUL(n)   "v_pk_mul_lo_u16 %" #n ", %4, %" #n
#define A_PKSUBS(n)  "v_pk_sub_u16 %" #n ", %4, %" #n " clamp"
#define A_ALIGNB(n)  "v_alignbit_b32 %" #n ", %4, %" #n ", %5"
#define A_ALIGNBY(n) "v_alignbyte_b32 %" #n ", %4, %" #n ", %5"
#define A_PERM(n)    "v_perm_b32 %" #n ", %4, %" #n ", %5"
#define A_BFE(n)     "v_bfe_u32 %" #n ", %" #n ", 3, 8"
#define A_LSHLOR(n)  "v_lshl_or_b32 %" #n ", %4, 16, %" #n
#define A_LSHLADD(n) "v_lshl_add_u32 %" #n ", %4, 2, %" #n
#define A_ANDOR(n)   "v_and_or_b32 %" #n ", %4, %5, %" #n
#define A_MIN3(n)    "v_min3_u32 %" #n ", %4, %5, %" #n
#define A_MAX3(n)    "v_max3_i32 %" #n ", %4, %5, %" #n
#define A_MIN(n)     "v_min_u32 %" #n ", %4, %" #n
#define A_SAD(n)     "v_sad_u8 %" #n ", %4, %5, %" #n
#define A_SAD16(n)   "v_sad_u16 %" #n ", %4, %5, %" #n
#define A_MSAD(n)    "v_msad_u8 %" #n ", %4, %5, %" #n
#define A_DOT4(n)    "v_dot4_u32_u8 %" #n ", %4, %5, %" #n
#define A_DOT4I(n)   "v_dot4_i32_i8 %" #n ", %4, %5, %" #n
#define A_DOT8(n)    "v_dot8_u32_u4 %" #n ", %4, %5, %" #n
#define A_DOT2(n)    "v_dot2_u32_u16 %" #n ", %4, %5, %" #n
#define A_BFI(n)     "v_bfi_b32 %" #n ", %4, %5, %" #n
#define A_XAD(n)     "v_xad_u32 %" #n ", %4, %5, %" #n
#define A_ADD3(n)    "v_add3_u32 %" #n ", %4, %5, %" #n
#define A_OR3(n)     "v_or3_b32 %" #n ", %4, %5, %" #n
#define A_MBCNT(n)   "v_mbcnt_lo_u32_b32 %" #n ", %4, %" #n
#define A_LSHR(n)    "v_lshrrev_b32 %" #n ", 3, %" #n
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %4, %" #n ", vcc"
#define A_FMA(n)     "v_fma_f32 %" #n ", %4, %5, %" #n
#define A_PKFMA(n)   "v_pk_fma_f16 %" #n ", %4, %5, %" #n
#define A_FFBH(n)    "v_ffbh_u32 %" #n ", %" #n
#define A_MADU16(n)  "v_mad_u16 %" #n ", %4, %5, %" #n
#define A_ADDSDWA(n) "v_add_u32_sdwa %" #n ", %4, %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
#define A_XORS(n)    "v_xor_b32 %" #n ", s4, %" #n
#define A_ANDLIT(n)  "v_and_b32 %" #n ", 0xff00ff, %" #n
#define A_ANDINL(n)  "v_and_b32 %" #n ", 63, %" #n
#define A_ANDV(n)    "v_and_b32 %" #n ", %4, %" #n
#define A_ORV(n)     "v_or_b32 %" #n ", %4, %" #n
#define A_SUBV(n)    "v_sub_u32 %" #n ", %4, %" #n
#define A_MAXU(n)    "v_max_u32 %" #n ", %4, %" #n
#define A_MAXU16(n)  "v_max_u16 %" #n ", %4, %" #n
#define A_ADDU16(n)  "v_add_u16 %" #n ", %4, %" #n
#define A_MULU16(n)  "v_mul_lo_u16 %" #n ", %4, %" #n
#define A_LSHLV(n)   "v_lshlrev_b32 %" #n ", %4, %" #n
#define A_ASHR(n)    "v_ashrrev_i32 %" #n ", 3, %" #n
#define A_MOV(n)     "v_mov_b32 %" #n ", %4"
#define A_CMPVCC(n)  "v_cmp_gt_u32 vcc, %4, %" #n
#define A_CMPS(n)    "v_cmp_gt_u32 s[20:21], %4, %" #n
#define A_CNDS(n)    "v_cndmask_b32 %" #n ", %4, %" #n ", s[20:21]"
#define A_CNDVCC2(n) "v_cndmask_b32 %" #n ", %4, %5, vcc"
#define A_ADDCO(n)   "v_add_co_u32 %" #n ", vcc, %4, %" #n
#define A_SUBREV(n)  "v_subrev_u32 %" #n ", %4, %" #n
#define A_CVT(n)     "v_cvt_f32_u32 %" #n ", %" #n
#define A_MULF(n)    "v_mul_f32 %" #n ", %4, %" #n
#define A_ADDF(n)    "v_add_f32 %" #n ", %4, %" #n
#define A_PKFMA32(n) "v_pk_fma_f32 %" #n ", %4, %5, %" #n   /* placeholder */
#define A_RCP(n)     "v_rcp_f32 %" #n ", %" #n
#define A_MOVDPP(n)  "v_mov_b32_dpp %" #n ", %4 row_shr:1 row_mask:0xf bank_mask:0xf"
#define A_ADDDPP(n)  "v_add_u32_dpp %" #n ", %4, %" #n " row_shr:1 row_mask:0xf bank_mask:0xf"
#define A_READLN(n)  "v_readlane_b32 s22, %" #n ", 3"
#define A_FMA64(n)   "v_add_f32 %" #n ", %4, %" #n
#define A_CND64VCC(n) "v_cndmask_b32_e64 %" #n ", %4, %" #n ", vcc"
#define A_CNDINIT(n)  "s_mov_b64 vcc, 0x5555\n v_cndmask_b32 %" #n ", %4, %" #n ", vcc"
#define A_CNDCONST(n) "v_cndmask_b32_e64 %" #n ", 0, 1, vcc"
#define A_CMPCND(n)   "v_cmp_gt_u32 vcc, %4, %" #n "\n v_cndmask_b32 %" #n ", %5, %" #n ", vcc"
#define A_LSHL64(n)  "v_lshlrev_b64 %" #n ", 3, %" #n   /* placeholder, not used */

DEFK(k_xor, A_XOR) DEFK(k_bcnt, A_BCNT) DEFK(k_add, A_ADD) DEFK(k_mul24, A_MUL24) DEFK(k_mad24, A_MAD24) DEFK(k_mullo, A_MULLO)
DEFK(k_pkadd, A_PKADD) DEFK(k_pkmin, A_PKMIN) DEFK(k_pkmad, A_PKMAD) DEFK(k_pkmul, A_PKMUL) DEFK(k_pksubs, A_PKSUBS)
DEFK(k_alignbit, A_ALIGNB) DEFK(k_alignbyte, A_ALIGNBY) DEFK(k_perm, A_PERM) DEFK(k_bfe, A_BFE) DEFK(k_lshlor, A_LSHLOR)
DEFK(k_lshladd, A_LSHLADD) DEFK(k_andor, A_ANDOR) DEFK(k_min3, A_MIN3) DEFK(k_max3, A_MAX3) DEFK(k_min, A_MIN)
DEFK(k_sad, A_SAD) DEFK(k_sad16, A_SAD16) DEFK(k_msad, A_MSAD) DEFK(k_dot4, A_DOT4) DEFK(k_dot4i, A_DOT4I) DEFK(k_dot8, A_DOT8) DEFK(k_dot2, A_DOT2)
DEFK(k_bfi, A_BFI) DEFK(k_xad, A_XAD) DEFK(k_add3, A_ADD3) DEFK(k_or3, A_OR3) DEFK(k_mbcnt, A_MBCNT) DEFK(k_lshr, A_LSHR)
DEFK(k_cndmask, A_CNDMASK) DEFK(k_fma, A_FMA) DEFK(k_pkfma16, A_PKFMA) DEFK(k_ffbh, A_FFBH) DEFK(k_madu16, A_MADU16) DEFK(k_addsdwa, A_ADDSDWA)
DEFK(k_xor_sgpr, A_XORS)
DEFK(k_andlit, A_ANDLIT) DEFK(k_andinl, A_ANDINL) DEFK(k_andv, A_ANDV) DEFK(k_orv, A_ORV) DEFK(k_subv, A_SUBV) DEFK(k_maxu, A_MAXU) DEFK(k_maxu16, A_MAXU16)
DEFK(k_addu16, A_ADDU16) DEFK(k_mulu16, A_MULU16) DEFK(k_lshlv, A_LSHLV) DEFK(k_ashr, A_ASHR) DEFK(k_mov, A_MOV) DEFK(k_cmpvcc, A_CMPVCC) DEFK(k_cmps, A_CMPS)
DEFK(k_cnds, A_CNDS) DEFK(k_cndvcc2, A_CNDVCC2) DEFK(k_addco, A_ADDCO) DEFK(k_subrev, A_SUBREV) DEFK(k_cvt, A_CVT) DEFK(k_mulf, A_MULF) DEFK(k_addf, A_ADDF) DEFK(k_rcp, A_RCP)
DEFK(k_cnd64vcc, A_CND64VCC) DEFK(k_cndinit, A_CNDINIT) DEFK(k_cndconst, A_CNDCONST) DEFK(k_cmpcnd, A_CMPCND) DEFK(k_movdpp, A_MOVDPP) DEFK(k_adddpp, A_ADDDPP) DEFK(k_readlane, A_READLN)

typedef void (*kern_t)(uint32_t*, int, uint32_t);
struct Entry { const char* name; kern_t k; };

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    uint32_t* out; hipMalloc(&out, 64);
    Entry es[] = {
        {"v_xor_b32", k_xor}, {"v_xor_b32 (sgpr src)", k_xor_sgpr}, {"v_add_u32", k_add}, {"v_bcnt_u32_b32", k_bcnt}, {"v_mbcnt_lo", k_mbcnt}, {"v_ffbh_u32", k_ffbh},
        {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24}, {"v_mul_lo_u32", k_mullo}, {"v_mad_u16", k_madu16},
        {"v_pk_add_u16", k_pkadd}, {"v_pk_min_u16", k_pkmin}, {"v_pk_sub_u16 clamp", k_pksubs}, {"v_pk_mul_lo_u16", k_pkmul}, {"v_pk_mad_u16", k_pkmad},
        {"v_alignbit_b32", k_alignbit}, {"v_alignbyte_b32", k_alignbyte}, {"v_perm_b32", k_perm}, {"v_bfe_u32", k_bfe}, {"v_bfi_b32", k_bfi},
        {"v_lshl_or_b32", k_lshlor}, {"v_lshl_add_u32", k_lshladd}, {"v_and_or_b32", k_andor}, {"v_or3_b32", k_or3}, {"v_add3_u32", k_add3}, {"v_xad_u32", k_xad},
        {"v_min_u32", k_min}, {"v_min3_u32", k_min3}, {"v_max3_i32", k_max3}, {"v_lshrrev_b32", k_lshr}, {"v_cndmask_b32", k_cndmask},
        {"v_sad_u8", k_sad}, {"v_sad_u16", k_sad16}, {"v_msad_u8", k_msad}, {"v_dot4_u32_u8", k_dot4}, {"v_dot4_i32_i8", k_dot4i}, {"v_dot8_u32_u4", k_dot8}, {"v_dot2_u32_u16", k_dot2},
        {"v_and_b32 literal", k_andlit}, {"v_and_b32 inline const", k_andinl}, {"v_and_b32 vgpr", k_andv}, {"v_or_b32", k_orv}, {"v_sub_u32", k_subv}, {"v_subrev_u32", k_subrev},
        {"v_max_u32", k_maxu}, {"v_max_u16", k_maxu16}, {"v_add_u16", k_addu16}, {"v_mul_lo_u16", k_mulu16}, {"v_lshlrev_b32 vgpr shift", k_lshlv}, {"v_ashrrev_i32", k_ashr}, {"v_mov_b32", k_mov},
        {"v_cmp_gt_u32 -> vcc", k_cmpvcc}, {"v_cmp_gt_u32 -> sgpr", k_cmps}, {"v_cndmask sgpr mask (VOP3)", k_cnds}, {"v_cndmask vcc 2 src", k_cndvcc2}, {"v_add_co_u32", k_addco}, {"v_cndmask e64 vcc", k_cnd64vcc}, {"s_mov vcc + v_cndmask e32", k_cndinit}, {"v_cndmask e64 0,1,vcc", k_cndconst}, {"v_cmp + v_cndmask (pair)", k_cmpcnd},
        {"v_cvt_f32_u32", k_cvt}, {"v_mul_f32", k_mulf}, {"v_add_f32", k_addf}, {"v_rcp_f32", k_rcp}, {"v_mov_b32 dpp row_shr", k_movdpp}, {"v_add_u32 dpp", k_adddpp}, {"v_readlane_b32", k_readlane},
        {"v_add_u32_sdwa", k_addsdwa}, {"v_fma_f32", k_fma}, {"v_pk_fma_f16", k_pkfma16},
    };
    const int iters = 2000, blocks = n_cu * 8;          // 8 x 4 waves per CU = 8 waves per SIMD
    const double instr_per_wave = (double)iters * 16 * 4;
    printf("device %s, %d CUs, %.0f MHz\n", prop.name, n_cu, mhz);
    for (auto& e : es) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        e.k<<<blocks, 256>>>(out, 10, 1);
        hipDeviceSynchronize();
        hipEventRecord(a);
        e.k<<<blocks, 256>>>(out, iters, 1);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        // every SIMD runs 8 waves x instr_per_wave instructions
        const double cyc = ms * 1e-3 * mhz * 1e6 / (8.0 * instr_per_wave);
        printf("%-24s %8.3f ms  %6.2f cycles / wave-instr / SIMD\n", e.name, ms, cyc);
    }
    return 0;
}
