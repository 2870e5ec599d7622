// libfat5.so -- C ABI over the gfx950 kernels (see include/fat5.h for the contract).
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <algorithm>
#include <atomic>

#include "../../include/fat5.h"
#include "attn_common.h"
#include "attn_launch.h"
#include "reduce_kernels.h"
#include "rowwise_kernels.h"
#include "adamw_kernels.h"
#include "fold_weights.h"

using namespace fat5;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  return fail(FAT5_EHIP, "%s: %s", what, hipGetErrorString(e));
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the 64-rows-per-wave forward takes over from this many waves of (64 rows x all keys) on (measured, tools/attn_time.py, (4,12,S,64):
// S = 768 -- 576 waves -- 16.2 / 19.4 us without / with the T5 bias against 18.6 / 21.7 us of the 32-row body, S = 1024: 19.7 / 23.3 against
// 22.4 / 26.1; at S = 512 -- 384 waves -- only its key-split variant (128-row workgroups: 192 of them) keeps up with the
// two-waves-per-32-rows split body of attn_fwd.h: 10.4 / 11.6 against 10.4 / 12.5 us)
constexpr long kFwd64MinWaves = 384;

// kernel-variant override of a call (fat5_attn_params.variant; tests / profilers): 1 forced on, 0 forced off, -1 library's choice
inline int vsel(int variant, int on_bit, int off_bit) { return (variant & on_bit) ? 1 : ((variant & off_bit) ? 0 : -1); }

struct BwdLayout {
  size_t delta_off, stat2_off, ds_off, drpe_off, scratch_off, total;
  bool kv64;        // dK/dV by the 64-keys-per-wave pipelined body (attn_bwd64.h); the dQ kernel then also writes its statistics
  bool kv64_half;   // ... in its half-length variant (128-key workgroups: two wave pairs, each half of the query steps)
  int kv64_mix_pf;  // > 0: both variants in one launch, this many (b, h) pairs per XCD as 256-key workgroups (attn_bwd_kv64_mixed_kernel)
  bool q64;         // dQ by the 64-rows-per-wave pipelined body (attn_bwd64.h)
  bool fused64;     // both 64-wide bodies in one launch when a call asks for both stages (attn_bwd_fused64_kernel); implies kv64 (256-key) and q64
  bool ds_staged;   // dense dS goes through the workspace and is reduced afterwards
  bool dbias_inkernel;  // dense (1, H, M, N) gradient by the batch-inner kernel (attn_bwd_dbias.h): nothing of size B*H*M*N
  bool dfused64;    // dense (1, H, M, N) bias: the 256-key dense dK/dV workgroups and the dQ + dBias ones in ONE launch behind bwd_stat2_kernel (attn_bwd_dfused64_kernel); implies kv64 (256-key) and qdb64
  bool qdb64;       // dense (1, H, M, N) bias shared by the batch: dQ AND the batch-reduced dbias by attn_bwd_qdb64_kernel (four batch elements per workgroup)
  int qdb_groups;   // ... ceil(B / 4); > 1: fp32 slabs in the workspace + dbias_partial_reduce_kernel
  int n_nblk;
  int nw_q, nw_kv;
  bool diag_q;     // T5 bias, one-launch 64-wide backward: the dQ workgroups form the partial diagonal sums (one partial row per 256-row block), the dK/dV ones none
  int part_rows;   // partial diagonal-sum rows per (b, h) in the workspace
};

// Compute units of the device the dispatch rules were measured on (MI355X: 256 in 8 XCDs) and of the device in use: the rules' workgroup-count
// thresholds are rounds of the chip, so they scale with its size.  Read once per device ordinal from hipDeviceProp (no hard-coded 256 / 32);
// without a device -- the host-only dispatch tests -- the reference chip is assumed.
static int chip_cus() {
  // per device (a process may drive GPUs of different sizes): the current device's count, read once per ordinal
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return 256;
  }
  const int slot = dev >= 0 && dev < 64 ? dev : 0;
  int n = cache[slot].load(std::memory_order_relaxed);
  if (n > 0) return n;
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, dev) != hipSuccess || pr.multiProcessorCount <= 0) {
    (void)hipGetLastError();
    return 256;
  }
  cache[slot].store(pr.multiProcessorCount, std::memory_order_relaxed);
  return pr.multiProcessorCount;
}
static inline long cu_scaled(long wgs_at_256) { return wgs_at_256 * chip_cus() / 256; }  // a workgroup-count threshold measured on 256 CUs

// head_dim 16 runs the D = 32 instantiations with AttnArgs::dvalid = 16: columns 16..31 are read as zeros (out-of-range DMA pieces, masked
// fragment loads) and never written -- no padded copies of q / k / v / do in HBM (the reference's kernels take 16 natively, flash_attention_v2_bias.py:233-234)
static inline int effD(const fat5_attn_params* p) { return p->D == 16 ? 32 : p->D; }

int pick_nw(long ctas_at_nw4) {
  // measured at S = 512: the 4-wave tile wins down to ~0.75 workgroups per CU
  return ctas_at_nw4 < cu_scaled(160) ? 2 : 4;
}

int check_common(const fat5_attn_params* p) {
  if (!p) return fail(FAT5_EINVAL, "params is NULL");
  if (p->D != 16 && p->D != 32 && p->D != 64 && p->D != 128) return fail(FAT5_EINVAL, "head_dim %d unsupported (16, 32, 64, 128)", p->D);
  if (p->dtype != FAT5_F16 && p->dtype != FAT5_BF16) return fail(FAT5_EINVAL, "dtype %d unsupported (f16, bf16)", p->dtype);
  if (p->B <= 0 || p->H <= 0 || p->M <= 0 || p->N <= 0) return fail(FAT5_EINVAL, "empty problem B=%d H=%d M=%d N=%d", p->B, p->H, p->M, p->N);
  if (p->bias_mode < 0 || p->bias_mode > 2) return fail(FAT5_EINVAL, "bias_mode %d", p->bias_mode);
  if (p->bias_mode == FAT5_BIAS_DENSE && !p->bias) return fail(FAT5_EINVAL, "dense bias mode without bias pointer");
  if (p->bias_mode == FAT5_BIAS_RPE1D) {
    if (!p->rpe1d) return fail(FAT5_EINVAL, "rpe1d mode without table");
    if (p->rpe_radius < 1 || p->rpe_radius > 2048) return fail(FAT5_EINVAL, "rpe_radius %d out of range [1, 2048]", p->rpe_radius);
  }
  if (p->unit_count < 0 || p->unit_begin < 0 || (p->unit_count > 0 && (long)p->unit_begin + p->unit_count > (long)p->B * p->H))
    return fail(FAT5_EINVAL, "unit range [%d, +%d) outside the %ld units of the problem", p->unit_begin, p->unit_count, (long)p->B * p->H);
  if (p->unit_count > 0 && p->cu_seqlens_q) return fail(FAT5_EINVAL, "unit ranges are not available for packed (cu_seqlens) batches");
  return FAT5_OK;
}
inline long n_units(const fat5_attn_params* p) { return p->unit_count > 0 ? p->unit_count : (long)p->B * p->H; }

bool strides_ok(const void* ptr, const int64_t* s) {
  return aligned16(ptr) && (s[0] % 8 == 0) && (s[1] % 8 == 0) && (s[2] % 8 == 0) && s[2] > 0;
}
// the kernels address one (b,h) slice through a 32-bit buffer descriptor: its rows must span < 2 GiB
bool slice_fits(int64_t rows, int64_t row_stride, int D) {
  return ((rows - 1) * row_stride + D) * 2 < (int64_t(1) << 31);
}

void fill_common(const fat5_attn_params* p, AttnArgs& a) {
  memset(&a, 0, sizeof(a));
  a.q = (const uint16_t*)p->q; a.k = (const uint16_t*)p->k; a.v = (const uint16_t*)p->v;
  a.o = (uint16_t*)p->o; a.lse = p->lse;
  for (int i = 0; i < 3; ++i) {
    a.qs[i] = p->q_stride[i]; a.ks[i] = p->k_stride[i]; a.vs[i] = p->v_stride[i]; a.os[i] = p->o_stride[i];
    a.bs[i] = p->bias_stride[i];
  }
  a.B = p->B; a.H = p->H; a.M = p->M; a.N = p->N;
  a.dvalid = p->D;
  // forward, pipelined sweep: with reference point 0 a row whose largest logit passes ~69 nats (2^100) sends its workgroup through the exact pass again.  The reference
  // benchmarks at sm_scale 1.3 on unit-variance inputs (benchmarks/bench_fa2_bias.py): logits of sigma 1.3 sqrt(D) = 10 .. 15 nats -- at d_head 128 a third of the
  // workgroups repeated (272 vs 130 us).  From sigma 8 on the sweep takes the rows' maxima over the first tile as reference point (a scores-only pre-pass of one tile).
  a.ref_first = std::fabs(p->sm_scale) * std::sqrt((float)p->D) >= 8.f;
  a.causal = p->causal; a.scale = p->sm_scale; a.R = p->rpe_radius;
  a.bias = (const uint16_t*)p->bias; a.rpe1d = p->rpe1d;
  a.cu_q = p->cu_seqlens_q; a.cu_k = p->cu_seqlens_k;
  a.total_q = p->total_q; a.total_k = p->total_k;
  a.unit_begin = p->unit_begin; a.unit_count = p->unit_count;
  // (pays once the bias no longer sits in the 256 MB Infinity Cache: measured +18 % forward at S = 8192 (1.6 GB), neutral at
  //  S = 2048 (100 MB), -10 % at S = 512 where the per-(b,h) mapping keeps K/V in one L2)
  a.batch_inner = p->bias_mode == FAT5_BIAS_DENSE && p->bias_stride[0] == 0 && p->B > 1 && p->unit_count == 0 &&
                  !p->cu_seqlens_q &&
                  (int64_t)(p->bias_stride[1] ? p->H : 1) * p->M * p->N * 2 > (int64_t(256) << 20);
  if (p->bias_mode == FAT5_BIAS_DENSE) {
    a.bias_vec4 = ((reinterpret_cast<uintptr_t>(p->bias) & 7) == 0) && (p->bias_stride[0] % 4 == 0) &&
                  (p->bias_stride[1] % 4 == 0) && (p->bias_stride[2] % 4 == 0);
    a.bias_dma = ((reinterpret_cast<uintptr_t>(p->bias) & 15) == 0) && (p->bias_stride[0] % 8 == 0) &&
                 (p->bias_stride[1] % 8 == 0) && (p->bias_stride[2] % 8 == 0);
  }
}

typedef hipError_t (*launch_fn)(const AttnArgs&, int, int, int, int, hipStream_t);

template <int V>
using IC = std::integral_constant<int, V>;
template <typename F>
void dispatch_dtype(int dt, F&& f) {
  switch (dt) {
    case FAT5_F32: f(IC<FAT5_F32>{}); break;
    case FAT5_F16: f(IC<FAT5_F16>{}); break;
    default: f(IC<FAT5_BF16>{}); break;
  }
}

}  // namespace

extern "C" {

int fat5_version(void) { return FAT5_VERSION; }
int fat5_chip_cus(void) { return chip_cus(); }
const char* fat5_last_error(void) { return g_err; }
size_t fat5_sizeof_attn_params(void) { return sizeof(fat5_attn_params); }

// Forward: which body runs a problem (waves per workgroup, split / pipelined bodies).  Chosen from the FULL problem -- the grid comes
// from the call's units: a unit-range call runs exactly the code the whole-problem call would run on those units, so sharded and
// unsharded results are bit-identical.
struct FwdChoice {
  int nw;        // launcher code: waves per workgroup of the 32-row body, -4 = its two-waves-per-32-rows split form; 64-row body: 4, 2 = key-split
  int n_mblk;    // query tiles per (b, h)
  bool fwd64;    // the 64-rows-per-wave pipelined body (attn_fwd64.h)
  bool ksplit;
  bool mixed;    // 256-row and key-split 128-row workgroups in one launch (attn_fwd64_mixed_kernel): mix_* filled in
  int mix_na, mix_a_lo, mix_k_hi;
  long mix_grid;
};
static FwdChoice fwd_choice(const fat5_attn_params* p) {
  FwdChoice c;
  c.fwd64 = c.ksplit = c.mixed = false;
  c.mix_na = c.mix_a_lo = c.mix_k_hi = 0;
  c.mix_grid = 0;
  const long bh = (long)p->B * p->H;
  int nw = pick_nw(bh * ((p->M + 127) / 128));
  c.n_mblk = (p->M + 32 * nw - 1) / (32 * nw);
  // short sequences: with 4-wave tiles the grid is smaller than the chip and every wave walks all keys alone -> two
  // waves per 32 query rows, each taking one 32-key block of every tile (attn_fwd_split_kernel)
  const long ctas4 = bh * ((p->M + 127) / 128);
  if (!(p->variant & FAT5_V_NO_SPLIT) && ctas4 <= chip_cus() && bh * ((p->M + 63) / 64) >= cu_scaled(96) && p->N >= 128) {
    nw = -4;
    c.n_mblk = (p->M + 63) / 64;
  }
  // long sequences: 64 query rows per wave, software-pipelined tile loop (attn_fwd64.h) once its 256-row workgroups fill
  // the chip (fat5_attn_params.variant: FAT5_V_FWD64_OFF disables, FAT5_V_FWD64_ON forces wherever the body applies)
  const int f64_env = vsel(p->variant, FAT5_V_FWD64_ON, FAT5_V_FWD64_OFF);
  const long waves64 = bh * ((p->M + 63) / 64);  // waves of 64 query rows x all keys
  const bool ctab = p->causal && p->bias_mode == FAT5_BIAS_RPE1D && p->N - p->M < p->rpe_radius && p->N - p->M >= -p->rpe_radius;  // (the bias table carries the causal mask: attn_fwd64.h)
  // Dense bias (round 4): the 64-row body with a two-tile bias ring in LDS -- one workgroup per CU, one wave per SIMD, so it needs
  // the chip full of 64-row waves; bf16 (the sweep without a running maximum), bias rows 16-byte aligned (LDS-DMA)
  const bool dense64 = p->D == 64 && p->bias_mode == FAT5_BIAS_DENSE && p->dtype == FAT5_BF16 && !p->cu_seqlens_q && f64_env != 0 &&
                       ((reinterpret_cast<uintptr_t>(p->bias) & 15) == 0) && (p->bias_stride[0] % 8 == 0) && (p->bias_stride[1] % 8 == 0) &&
                       (p->bias_stride[2] % 8 == 0) && smem_fwd64_d64(0, FAT5_BIAS_DENSE) <= 160 * 1024 &&
                       (f64_env == 1 || (!p->causal && ((waves64 >= cu_scaled(3072) && p->N >= 4096) || (waves64 >= cu_scaled(6144) && p->N >= 2048))));  // (round 5: (16,12,2048) 324 vs 360 us)  // (measured, us, 64-row vs 32-row body: (4,12,8192) 1217 vs 1371; (4,12,2048) 102 vs 95; (16,12,1024) causal 95 vs 77)
  // ... head_dim 128 (round 5): the one-wave-per-SIMD form of the body with three ring slots beside the bias ring (160 KB)
  const bool dense128 = p->D == 128 && p->bias_mode == FAT5_BIAS_DENSE && p->dtype == FAT5_BF16 && !p->cu_seqlens_q && f64_env != 0 &&
                        ((reinterpret_cast<uintptr_t>(p->bias) & 15) == 0) && (p->bias_stride[0] % 8 == 0) && (p->bias_stride[1] % 8 == 0) &&
                        (p->bias_stride[2] % 8 == 0) && smem_fwd64_d128(0, FAT5_BIAS_DENSE) <= 160 * 1024 &&
                        // (measured, us, 64-row vs 32-row body: (16,12,1024) 127 vs 244, causal 130 vs 213; (4,12,8192) 1872 vs 3734; (2,12,1024) 39.6 vs 48.8, (1,12,2048) 68.8 vs 83.4,
                        //  (4,12,1024) causal 45.7 vs 66.9; (1,12,1024) -- 192 waves -- 38.2 vs 26.4)
                        //  (16,12,512) 49.0 vs 74.9, causal 58.3 vs 75.8)
                        //  (audit, profiles/r05_dispatch_audit_d128_fwd.log: (8,12,512) 27.7 vs 49.5, causal 28.6 vs 50.5; (4,12,512) 25.5 vs 28.5 -> from 512 keys at every size;
                        //   causal below 768 waves and 2048 keys the 32-row body stays: (2,12,1024) causal 41.1 vs 36.9, (4,12,512) causal 23.5 either way)
                        (f64_env == 1 || (waves64 >= cu_scaled(384) && p->N >= 512 && !(p->causal && waves64 < cu_scaled(768) && p->N < 2048)));
  if (dense64 || dense128) {
    c.fwd64 = true;
    c.ksplit = false;
    c.nw = 4;
    c.n_mblk = (p->M + 255) / 256;
    return c;
  }
  // head_dim 128 (round 5): the same body at ONE wave per SIMD (O^T alone is 128 registers per lane), 32 MFMA gaps per block, 256-row workgroups;
  // bias none / rpe1d.  Needs the chip full of 64-row waves.
  if (p->D == 128 && p->bias_mode != FAT5_BIAS_DENSE && !p->cu_seqlens_q && f64_env != 0 &&
      smem_fwd64_d128(p->rpe_radius, p->bias_mode) <= 160 * 1024 &&
      // (measured, us, 64-row vs 32-row body, tools/attn_time.py --D 128: (4,12,1024) 30.6 vs 39.8, (2,12,2048) 51.3 vs 63.5, (1,12,4096) 94.8 vs 120.6, (16,12,4096) 1305 vs 1628,
      //  (4,12,8192) 1265; below 768 waves it loses -- (2,12,1024) 27.3 vs 22.4 -- and at 512 keys it ties: (16,12,512) 39.7 vs 39.8.  Plain causal (diagonal
      //  tiles unpipelined): (16,12,1024) 105 vs 93, (4,12,2048) 87.5 vs 88.2, (16,12,4096) 843 vs 893; with the T5 table carrying the mask: (16,12,1024) 88.7 vs 96.6)
      //  (masked blocks inside the pipelined sweep, round 5: plain causal (16,12,1024) 92.3 vs 92.6, (4,12,1024) 32.2 vs 38.5, (4,12,4096) 237 vs 259)
      //  (audit, profiles/r05_dispatch_audit_d128_fwd.log: at 512 keys the body wins where its workgroups are one partial round -- (8,12,512) none 21.6 vs 24.9, T5 table 21.9 vs 26.6,
      //   causal 22.9 vs 25.9 / 21.7 vs 25.6 -- and ties at 1.5 rounds: (16,12,512) 39.6 / 42.4 either way)
      (f64_env == 1 || (waves64 >= cu_scaled(768) && (p->N >= 1024 || (p->N >= 512 && waves64 <= 4L * chip_cus()))))) {
    c.fwd64 = true;
    c.n_mblk = (p->M + 255) / 256;
    c.nw = bh * c.n_mblk <= chip_cus() ? 5 : 4;  // (5: ring requests spread over the MFMA gaps -- one partial round, every CU in the same phase: (4,12,1024) 32.4 vs 38.7 us)
    return c;
  }
  if (p->D == 64 && p->bias_mode != FAT5_BIAS_DENSE && !p->cu_seqlens_q && f64_env != 0 &&
      // (fp16, round 4: the pipelined sweep with the first tile's row maxima as reference point -- (4,12,8192) 739 us against 913 for the
      //  32-row body, (4,12,2048) 64.3 vs 69.4)
      (f64_env == 1 || (waves64 >= cu_scaled(p->dtype == FAT5_BF16 ? kFwd64MinWaves : 1536) &&
                        // (512 keys are 8 tiles: too few for the pipeline's prologue to pay when half of them sit on the causal diagonal --
                        //  tools/dispatch_audit.py: (4,12,512) causal 10.0 vs 11.2 us, (8,12,512) causal 15.7 vs 17.5 -- or in the 1.5-waves-per-SIMD
                        //  range where the 64-row waves fill the chip unevenly: (16,12,512) 22.2 vs 24.8; (16,12,1024x512) and (16,12,2048x512),
                        //  2 and 4 full rounds, stay on the 64-row body: 33.6 vs 36.8, 62.4 vs 68.5)
                        // (causal with the T5 bias and the diagonal inside the band -- ctab, round 4: the table carries the mask, diagonal tiles are pipelined band
                        //  tiles: (4,12,512) causal 9.9 vs 10.8 us, (8,12,512) 13.7 vs 16.9, (16,12,512) 21.7 vs 23.8)
                        //  (round-4 audit, profiles/r04_dispatch_audit.log: the non-causal 1.5-waves-per-SIMD exception at <= 512 keys is gone -- (16,12,512) T5 bias 23.3
                        //   key-split vs 25.0 us for the 32-row body, 21.7 vs 22.0 without bias)
                        !(p->N <= 512 && p->causal && !ctab) &&
                        // (closing audit of round 5, profiles/r05c_dispatch_audit_H12.log: (2,12,1024) causal, 384 waves -- 15.5 vs 12.5 us for the 32-row body, T5 bias 14.8 vs 13.4;
                        //  (4,12,1024) causal, 768 waves, 15.6 either way -> causal problems of 513 .. 2047 keys from 768 waves on)
                        // (off-grid audit of round 6: (5,12,768) causal, 720 waves -- 13.9 vs 16.5 us for the 32-row body, T5 bias 13.4 vs 19.5 -> from 512 waves on)
                        !(p->causal && p->N > 512 && p->N < 2048 && waves64 < cu_scaled(512)) &&
                        // (its two waves per SIMD need two workgroups per CU: a radius beyond ~500 takes the table past 80 KB of LDS)
                        smem_fwd64_d64(p->rpe_radius, p->bias_mode) <= 80 * 1024))) {
    c.fwd64 = true;
    // Key-split variant (two waves per 64 rows, 128-row workgroups): where the 64-row waves fill between one and two slots of the
    // chip's 1024 SIMDs -- half the SIMDs then carry two full-length waves and the others one.  Measured at (4,12,2048,64) (tools/
    // attn_time.py): 60.7 vs 63.8 us with the T5 bias (a 128-row workgroup also crosses fewer band tiles), 58.4 vs 59.1 without;
    // slower everywhere else (S = 1024: 22.0 vs 19.7 us, 4096: 207 vs 190, 8192: 794 vs 724).  The hardware packs the workgroups of the
    // last, half-empty round two to a CU, so the finer grain buys less than a per-SIMD issue model predicts.
    const int ks_env = vsel(p->variant, FAT5_V_FWD64_KSPLIT_ON, FAT5_V_FWD64_KSPLIT_OFF);
    // Causal: the workgroups of a (b, h) pair have unequal lengths, finer units balance better -- (4,12,1024) 22.0 vs 23.5 us, (8,12,2048)
    // 82 vs 91, (4,12,4096) 137-140 vs 147-149, (8,12,4096) 245 vs 250, equal from ~8000 waves on (tools/dispatch_audit.py).
    // (only where the mask actually shortens workgroups: with N >= 2 M every row sees most keys -- (16,12,1024x4096) causal 181 vs 196 us plain)
    // (round-4 audit: also between 512 and 1024 waves -- (4,12,1024) T5 bias 20.7 vs 22.6 us, none 19.4 vs 19.9; (8,12,512) 14.3 vs 16.1, 12.9 vs 13.5)
    // (round-6 audit, profiles/r06_dispatch_audit_H12.log: since round 5 -- causal launches longest-first, masked blocks inside the pipelined sweep -- the 256-row form
    //  wins on causal problems from 3072 waves on: (8,12,2048) causal 66.3 split vs 64.7 us, T5 bias 66.7 vs 64.3; (4,12,4096) 114.2 vs 110.8 / 116.6 vs 108.4;
    //  (16,12,2048) 136.6 vs 124.4 / 136.0 vs 122.8; (8,12,4096) 230.4 vs 207.7 / 226.6 vs 204.0; (4,12,8192) 406.5 vs 373.4; (16,12,1024) T5 bias 45.8 vs 43.2 --
    //  the causal extension to 8192 waves of rounds 3-4 is gone; below 2048 waves the split form stays for every mask)
    //  (B H = 64 audit, profiles/r06_dispatch_audit_other_H.log: at exactly 2048 waves the split form still wins on causal problems -- (2,32,2048) causal 46.2 vs 49.0 us,
    //   (4,16,2048) 46.6 vs 48.5, (8,8,2048) 46.2 vs 48.4 -- and ties on full ones (69.5 either way) -> causal: below 2560 waves)
    const bool ksplit = ks_env == 1 || (ks_env != 0 && waves64 < cu_scaled(p->causal ? 2560 : 2048));
    // (round-4 audit: without bias as well -- (16,12,1024) causal 51.1 split vs 55.1 us, (16,12,2048) 138.7 vs 151.5: the split form's Q / O now travel as whole rows)
    // (ctab: (16,12,1024) causal 50.6 split vs 53.4, (16,12,2048) 140 vs 147 -- the diagonal tiles no longer cost the split form an exact tile each)
    // (... and only while a wave still has keys to split: (16,12,1024) causal, 3072 waves of 16 tiles, 55.4 us plain vs 57.6 split;
    //  interleaved A/B timings: tools/ab_variants.py)
    c.ksplit = ksplit;
    nw = ksplit ? 2 : 4;
    c.n_mblk = ksplit ? (p->M + 127) / 128 : (p->M + 255) / 256;
    // Both forms in one launch (round 4): between one and two 64-row waves per SIMD either pure form leaves half of the SIMDs with
    // twice the work of the others; one 256-row workgroup (a full-length wave per SIMD) plus one key-split 128-row workgroup (a
    // half-length wave per SIMD) per CU gives every SIMD 1.5 units.  n = waves64 / 6 workgroups of each kind: taken where that is one
    // round of the chip (0.9 .. 1.1 x 256), the pairs divide evenly over the XCDs and a pair has room for a 256-row workgroup.
    // Measured (tools/attn_time.py, us, mixed vs the key-split form): (4,12,2048) T5 bias 54.4-55.8 vs 58.0-58.4, none 52.6-53.7 vs 56.7; (8,12,1024) 31.7-32.5 vs
    // 33.8; (2,12,4096) 101.2 vs 108.0.  The per-SIMD model promised -40 %: traced (HW_ID per workgroup), every CU does get one workgroup of each
    // kind, but the half-length waves -- one block per tile barrier -- take as long as the full-length ones beside them (86 k vs 77 k cycles).
    const int mx_env = vsel(p->variant, FAT5_V_FWD64_MIX_ON, FAT5_V_FWD64_MIX_OFF);
    if (mx_env != 0 && bh % 8 == 0 && p->M >= 384 && !p->causal && ks_env == -1 &&
        (mx_env == 1 || (waves64 * 10 >= 6L * chip_cus() * 9 && waves64 * 10 <= 6L * chip_cus() * 11 && p->N >= 1024))) {
      const long npx = bh / 8;                                   // pairs per XCD
      long na_x = (waves64 + 24) / 48;                           // 256-row workgroups per XCD (waves64 / 6 in total)
      na_x = std::max(na_x, npx);                                // (at least one per pair)
      const long a_max = (p->M - 128) / 256;                     // (a pair keeps at least one 128-row workgroup)
      na_x = std::min(na_x, npx * a_max);
      if (a_max >= 1) {
        c.mixed = true;
        c.mix_a_lo = (int)(na_x / npx);
        c.mix_k_hi = (int)(na_x % npx);
        c.mix_na = (int)(8 * na_x);
        const long b_hi = (std::max<long>(p->M - 256L * (c.mix_a_lo + 1), 0) + 127) / 128, b_lo = (std::max<long>(p->M - 256L * c.mix_a_lo, 0) + 127) / 128;
        c.mix_grid = p->unit_count > 0 ? (long)p->unit_count * (c.mix_a_lo + 1 + b_lo)   // (a unit range: fixed slots per pair, see the kernel)
                                       : 8 * (na_x + c.mix_k_hi * b_hi + (npx - c.mix_k_hi) * b_lo);
        nw = 3;
      }
    }
  }
  c.nw = nw;
  return c;
}

int fat5_attn_fwd(const fat5_attn_params* p, void* stream_) {
  int rc = check_common(p);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  if (!p->q || !p->k || !p->v || !p->o || !p->lse) return fail(FAT5_EINVAL, "fwd: null tensor pointer");
  if (!strides_ok(p->q, p->q_stride) || !strides_ok(p->k, p->k_stride) || !strides_ok(p->v, p->v_stride) ||
      !strides_ok(p->o, p->o_stride))
    return fail(FAT5_EINVAL, "fwd: q/k/v/o must be 16-byte aligned with strides that are multiples of 8 elements");
  if (!slice_fits(p->N, p->k_stride[2], p->D) || !slice_fits(p->N, p->v_stride[2], p->D) || !slice_fits(p->M, p->q_stride[2], p->D))
    return fail(FAT5_EINVAL, "fwd: one (batch, head) slice must span less than 2 GiB");
  if ((p->cu_seqlens_q == nullptr) != (p->cu_seqlens_k == nullptr)) return fail(FAT5_EINVAL, "cu_seqlens_q/k must both be set");
  if (p->cu_seqlens_q && p->bias_mode == FAT5_BIAS_DENSE) return fail(FAT5_EINVAL, "varlen: dense bias unsupported (none or rpe1d)");

  AttnArgs a;
  fill_common(p, a);
  const FwdChoice fc = fwd_choice(p);
  int nw = fc.nw;
  a.n_mblk = fc.n_mblk;
  a.mix_na = fc.mix_na; a.mix_a_lo = fc.mix_a_lo; a.mix_k_hi = fc.mix_k_hi;
  launch_fn fn = fc.fwd64 ? (p->D == 128 ? launch_fwd64_d128 : launch_fwd64_d64) : (effD(p) == 32 ? launch_fwd_d32 : (p->D == 64 ? launch_fwd_d64 : launch_fwd_d128));
  const long grid = fc.mixed ? fc.mix_grid : n_units(p) * a.n_mblk;
  if (grid > 0x7fffffffL) return fail(FAT5_EINVAL, "grid too large");
  hipError_t e = fn(a, p->dtype == FAT5_BF16, p->bias_mode, nw, (int)grid, stream);
  if (e != hipSuccess) return hip_fail(e, "attn_fwd launch");
  return FAT5_OK;
}

static int bwd_layout_compute(const fat5_attn_params* p, BwdLayout& L) {
  const long bh = (long)p->B * p->H;  // workspace layout and kernel variants follow the full problem also for a unit range
  L.nw_q = pick_nw(bh * ((p->M + 127) / 128));
  L.nw_kv = pick_nw(bh * ((p->N + 127) / 128));
  L.n_nblk = (p->N + 32 * L.nw_kv - 1) / (32 * L.nw_kv);
  // long sequences: 64 keys per wave, software-pipelined (attn_bwd64.h) once its 256-key workgroups (one per CU) cover the
  // chip twice (variant: FAT5_V_KV64_OFF disables, FAT5_V_KV64_ON forces wherever the body applies; FAT5_V_Q64_* likewise)
  const int b64_env = vsel(p->variant, FAT5_V_KV64_ON, FAT5_V_KV64_OFF);
  // One workgroup per CU (one wave per SIMD): `wg256` 256-key workgroups take ceil(wg256 / 256) rounds.  The half-length variant
  // (twice the workgroups, half the steps each) where that wastes more than a quarter of the last round, e.g. (4,12,2048,64):
  // 384 workgroups = 2 rounds against 768 half-length ones = 3 half rounds.  Measured (tools/attn_time.py, dK/dV stage, us):
  //   S = 2048  no bias: 32-key body 118, 256-key 112, half-length 108.7     T5 bias: 135.9 / 155.4 / 152.4
  //   S = 3072  no bias: 256-key 247, half-length 235                        T5 bias: 307.6 / 311.0
  // -- five DMA pieces per wave and step instead of three and a third prologue eat most of the round model's 25 %, and with the
  // T5 bias the ~12 band steps of a wave (general iteration + skew-tile sums, ~2.9x a pipelined step) decide: below 512 workgroups
  // the 64-key bodies take over only without bias.
  const long wg256 = bh * ((p->N + 255) / 256);
  const int kvh_env = vsel(p->variant, FAT5_V_KV64_HALF_ON, FAT5_V_KV64_HALF_OFF);
  const int mix_env = vsel(p->variant, FAT5_V_KV64_MIX_ON, FAT5_V_KV64_MIX_OFF);
  const long ncu = chip_cus();
  const double r_full = (double)((wg256 + ncu - 1) / ncu), r_half = 0.5 * 1.04 * (double)((2 * wg256 + ncu - 1) / ncu);
  // dense bias (round 5): the 256-key form of the 64-key body with the step's bias tile as one more LDS image per wave (attn_bwd64.h, DENSE): bf16,
  // bias rows 16-byte aligned (LDS-DMA)
  const bool dense = p->bias_mode == FAT5_BIAS_DENSE;
  // (the bodies add bias / scale on the matrix pipe, 1 / scale as two 16-bit terms: a zero scale keeps the older bodies)
  // (fp16: 1 / scale itself has to be an fp16 value in range -- |scale| down to 2e-5)
  const bool scale_exact = p->sm_scale != 0.f && std::isfinite(1.f / p->sm_scale) && std::isfinite(p->sm_scale) && (p->dtype == FAT5_BF16 || std::fabs(1.f / p->sm_scale) <= 60000.f);
  const bool dense_kv_ok = dense && scale_exact && ((reinterpret_cast<uintptr_t>(p->bias) & 15) == 0) && (p->bias_stride[0] % 8 == 0) &&
                           (p->bias_stride[1] % 8 == 0) && (p->bias_stride[2] % 8 == 0) &&
                           ((int64_t)(p->M - 1) * p->bias_stride[2] + p->N) * 2 < (int64_t(1) << 31);
  // Dense (1, H, M, N) bias shared by the batch (the reference's own operator): is the dQ + dBias body (attn_bwd_qdb64.h) legal for this call?
  const int qdb_env = vsel(p->variant, FAT5_V_QDB64_ON, FAT5_V_QDB64_OFF);
  const int qdb_ngrp = (p->B + 3) / 4;
  const bool qdb_legal = dense && p->dbias && p->D == 64 &&
                         scale_exact && p->dbias_batch == 1 && p->dbias_heads == p->H && p->bias_stride[0] == 0 && (p->bias_stride[1] != 0 || p->H == 1) &&
                         p->unit_count == 0 && !p->cu_seqlens_q && p->N % 8 == 0 && ((reinterpret_cast<uintptr_t>(p->bias) & 15) == 0) &&
                         (p->bias_stride[1] % 8 == 0) && (p->bias_stride[2] % 8 == 0) && (p->B > 1 || p->bias_stride[0] == 0) &&
                         (int64_t)p->M * p->N * (qdb_ngrp > 1 ? 4 : 2) < (int64_t(1) << 31) &&
                         // (B > 4: one fp32 (H, M, N) slab per group of four batch elements in the workspace -- (16,12,8192) would be 12.9 GB where the batch-inner kernel it
                         //  replaces needs one 3.2 GB buffer: beyond 4 GiB of slabs the older paths keep the call)
                         (qdb_ngrp == 1 || (int64_t)qdb_ngrp * p->H * p->M * p->N * 4 <= (int64_t(4) << 30)) &&
                         ((int64_t)(p->M - 1) * p->bias_stride[2] + p->N) * 2 < (int64_t(1) << 31);
  // (measured, whole dense backward, both 64-wide bodies against both older ones, us -- profiles/r05_dfused_time2.log: (4,12,768) 60.5 vs 78.5, (8,12,512) 55.1 vs 72.8, (2,12,1024) 71.9 vs 83.5,
  //  (16,12,256) 39.6 vs 47.6; (8,12,256) 36.6 vs 28.9, (4,12,256) 29.8 vs 29.5 -> from 2^23 scores per call on; below that only inside the one-launch form, see dfused64)
  // (closing audit of round 6, the one dense row above 5 %: (8,12,512) CAUSAL -- 192 + 192 workgroups of unequal length, one and a half rounds of the chip whether as one launch
  //  or two -- 58.9 us against 51.1 for the 32-wide one-launch body + staged dS; one round ((4,12,512): 31.9 vs 41.8) and two rounds ((16,12,512): 81.7 vs 92.6) stay)
  const long qdb_nq = (long)p->H * qdb_ngrp * ((p->M + 63) / 64);
  const bool dense_causal_odd_round = dense && p->causal && p->N <= 512 && wg256 + qdb_nq > chip_cus() && wg256 + qdb_nq < 2L * chip_cus();
  // (the same audit off the power-of-two grid, profiles/r06_audit_dense_small.log, whole dense backward, us: (6,12,384) 54.1 vs 37.2 for the 32-wide one-launch body + staged dS,
  //  (5,12,384) 53.8 vs 37.7, (7,12,384) 53.4 vs 42.4, (8,12,384) 54.5 vs 49.6, (5,12,512) 54.7 vs 48.8; (6,12,512) 55.3 vs 59.7, (7,12,512) 55.6 vs 62.7, (8,12,512) 56.4 vs 69.0:
  //  the two 64-wide launches cost ~54 us whatever the size down there -> from 2^24 scores on; 2^23 .. 2^24 only for the many short workgroups of B >= 16 ((16,12,256) 39.6 vs 47.6))
  const int64_t dense_scores = (int64_t)p->B * p->H * p->M * p->N;
  const bool qdb_rule = p->B >= 2 && (dense_scores >= (int64_t(1) << 24) || (dense_scores >= (int64_t(1) << 23) && p->B >= 16)) && !dense_causal_odd_round;
  const bool qdb_pick = qdb_legal && dense_kv_ok && qdb_env != 0 && qdb_rule && !(p->variant & (FAT5_V_DBIAS_STAGED | FAT5_V_DBIAS_INKERNEL));
  // (causal, closing audit of round 6 -- profiles/r06_audit_causal_kv.log: with longest-first launches the pure 256-key form beats the half-length and the mixed one on every
  //  causal problem measured, (8,12,1024) 41.7 vs 47.4 / 52.7 us, (5,12,1536) 48.1 vs 57.4 / 57.4, (16,12,1536) 141.3 vs 168.0 / 155.2 -> causal: only when a call forces them)
  L.kv64_half = !dense && (kvh_env == 1 || (kvh_env != 0 && p->bias_mode == FAT5_BIAS_NONE && r_half < 0.9 * r_full && !p->causal));
  // Both variants in one launch (attn_bwd_kv64_mixed_kernel): the first `pf` (b, h) pairs of every XCD as 256-key workgroups, the
  // others half-length.  pf by a list-scheduling model of one XCD (32 CUs, one workgroup per CU, launch order; a half-length
  // workgroup costs 0.65 of a full one without bias, 0.73 with the T5 bias: measured round times 56 / 36 us and 70 / 51 us at
  // S = 2048); taken when it beats both pure variants by 5 %.
  L.kv64_mix_pf = -1;
  double mix_gain = 1.0;
  if (bh % 8 == 0 && mix_env != 0 && kvh_env == -1 && !dense) {
    const int per = (int)(bh / 8), ntf = (p->N + 255) / 256, nth = (p->N + 127) / 128;
    const double rh = p->bias_mode == FAT5_BIAS_NONE ? 0.65 : 0.73;
    // Greedy list scheduling with two job sizes, without walking the jobs (this runs on the host inside every backward call): the F unit
    // jobs leave `rem` CUs at load a + 1 and the others at a; the k-th half-length job of a CU starts at load + (k - 1) * rh, greedy fills
    // these start slots in increasing order, so the makespan is (Hn-th smallest start) + rh -- O(Hn / 32) steps per candidate.
    auto makespan = [&](int pf) {
      const long F = (long)pf * ntf, Hn = (long)(per - pf) * nth;
      const long cx = chip_cus() / 8 > 0 ? chip_cus() / 8 : 1;  // CUs of one XCD
      const long a = F / cx, rem = F % cx, n_lo = cx - rem, n_hi = rem;
      const double base = (double)(a + (rem ? 1 : 0));
      if (Hn == 0) return base;
      long jl = 0, jh = 0, count = 0;
      double t = 0;
      while (count < Hn) {
        const double tl = (double)a + jl * rh, th = (double)(a + 1) + jh * rh;
        if (n_hi == 0 || tl <= th) { t = tl; count += n_lo; ++jl; }
        else { t = th; count += n_hi; ++jh; }
      }
      return std::max(t + rh, base);
    };
    {
      const double pure = std::min(makespan(per), L.kv64_half || p->bias_mode == FAT5_BIAS_NONE ? makespan(0) : 1e30);
      double best = 1e30;
      int best_pf = -1;
      for (int pf = 1; pf < per; ++pf) {
        const double m = makespan(pf);
        if (m < best) { best = m; best_pf = pf; }
      }
      // (causal: workgroups of unequal length -- the finer units of a mixed launch balance better than the uniform model says:
      //  (4,12,4096) causal 218 vs 233 us, T5 bias 271 vs 285; (8,12,4096) 432 vs 445 / 526 vs 549)
      // (round 6, after causal launches went longest-first in round 5 -- profiles/r06_audit_s4096.log: the pure 256-key launch now wins on causal problems,
      //  (8,12,3072) causal 236.0 vs 255.4 us mixed, T5 bias 260.4 vs 280.2; (8,12,4096) 395.6 vs 425.2 / 433.4 vs 460.2; (4,12,6144) 426.9 vs 446.0 / 463.5 vs 481.7:
      //  the mixed launch deals pair-major -> a causal problem takes it only where the model asks for it, like every other)
      // (closing audit: where the model did ask -- (8,12,1024) causal, 384 workgroups -- the pure 256-key launch was the fastest forced form (41.3 us): the model
      //  assumes workgroups of equal length -> causal problems take the mixed launch when a call forces it, never by the model)
      if (best_pf > 0 && (mix_env == 1 || (best < 0.95 * pure && !p->causal))) {
        L.kv64_mix_pf = best_pf;
        mix_gain = best / makespan(per);
      }
    }
  }
  const size_t kv64_lds = L.kv64_mix_pf > 0 ? std::max(smem_bwd_kv64h_d64(p->rpe_radius, p->bias_mode), smem_bwd_kv64_d64(p->rpe_radius, p->bias_mode))
                                            : (L.kv64_half ? smem_bwd_kv64h_d64(p->rpe_radius, p->bias_mode) : smem_bwd_kv64_d64(p->rpe_radius, p->bias_mode));
  // (the 64-key bodies take over from the 32-key one at 512 workgroups of 256 keys; from 320 where the last round is filled by
  //  half-length workgroups -- the mixed launch, or without bias the pure half-length variant)
  const bool fills = L.kv64_mix_pf > 0 ? mix_gain <= 0.9 : L.kv64_half;
  // Causal: the steps of a workgroup that touch the diagonal (256 keys / 32 rows = 8 of them) run the general, unpipelined iteration
  // at ~2.5x the cost of a pipelined step -- half of all steps at S = 1024, 6 % at 8192.  Measured dK/dV, 64-key against 32-key body:
  // (4,12,8192) 853 vs 940 us, (4,12,4096) 239 vs 246, (4,12,2048) T5 bias 118 vs 104, (16,12,2048) 298 vs 273, (16,12,1024) 111 vs 95
  // -> causal problems take the 64-key body from 4096 keys on.
  // (round 4: with the T5 bias and the diagonal inside the band the table carries the mask and the diagonal steps are pipelined band steps: (16,12,2048) causal
  //  64-key mixed 292 vs 326 us, (8,12,2048) 154 vs 169 -> from 2048 keys on)
  const bool ctab_kv = p->causal && p->bias_mode == FAT5_BIAS_RPE1D && p->N - p->M < p->rpe_radius && p->N - p->M >= -p->rpe_radius;
  // (round 5: without bias the mask rides in the score MFMAs' C operand -- (4,12,2048) causal 73.7 (half-length) vs 82.7 us, (16,12,2048) 261 vs 269;
  //  (16,12,1024) 102 vs 95 -> from 2048 keys on as well; dense: see dense_rule)
  // (round-6 audit at B = 16, profiles/r06_dispatch_audit_16x12.log -- with causal launches longest-first: (16,12,1024) causal 74.4 vs 80.2 us for the 32-key body,
  //  T5 bias 84.9 vs 97.7; (8,12,1024) 41.3 vs 43.8 -> from 1024 keys on)
  const bool kv64_causal_ok = !p->causal || p->N >= ((ctab_kv || p->bias_mode == FAT5_BIAS_NONE) ? 1024 : 4096);
  // (dense, round 5 -- bias on the matrix pipe, causal mask in the C operand; 64-key vs 32-key body, us: (4,12,2048) 137 vs 164, (4,12,8192) 1700 vs 2208;
  //  causal (16,12,512) 52.5 vs 54.1, (16,12,1024) 121 vs 133, (16,12,2048) 344 vs 392 -> from 192 workgroups on)
  //  (with the dQ + dBias body taken -- qdb_pick above -- the 64-key body follows it down to 2^23 scores)
  const bool dense_rule = dense && ((wg256 >= cu_scaled(192) && (int64_t)bh * p->M * p->N >= (int64_t(1) << 25)) || qdb_pick);
  L.kv64 = p->D == 64 && (!dense || dense_kv_ok) && !p->cu_seqlens_q && b64_env != 0 &&
           (b64_env == 1 || dense_rule || ((wg256 >= cu_scaled(p->causal ? 256 : (fills ? 320 : 512)) ||  // (causal: unequal workgroups fill the last round by themselves -- (8,12,1024) causal, 384 workgroups, 41.7 vs 43.7 us for the 32-key body; (5,12,1536), 360: 48.1 vs 53.0; (6,12,1024), 288: 33.4 vs 34.1)
                              // (long query streams pay even with the chip under-filled: 192 workgroups at (4,12,4096x1024) 99.5 vs 121.7 us,
                              //  (4,12,8192x1024) 190 vs 234; with the T5 bias 117 vs 146 and 213 vs 274)
                              (wg256 >= cu_scaled(160) && p->M >= 4096 && !p->causal)) && kv64_causal_ok)) && kv64_lds <= 160 * 1024;
  if (L.kv64 && L.kv64_mix_pf > 0) {
    L.kv64_half = false;
    L.nw_kv = 3;  // (launch_bwd_kv64: 3 selects the mixed launch)
    L.n_nblk = (p->N + 127) / 128;
  } else if (L.kv64) {
    L.kv64_mix_pf = -1;
    L.nw_kv = L.kv64_half ? 2 : 4;  // (launch_bwd_kv64: 2 selects the half-length variant)
    L.n_nblk = L.kv64_half ? (p->N + 127) / 128 : (p->N + 255) / 256;
  } else {
    L.kv64_half = false;
    L.kv64_mix_pf = -1;
  }
  const int q64_env = vsel(p->variant, FAT5_V_Q64_ON, FAT5_V_Q64_OFF);
  // (the 64-row dQ body: its prologue and the diagonal steps of a causal problem need long key streams to pay -- measured against the
  //  32-row body: non-causal (16,12,1024) 92 vs 85 us, (4,12,8192) 1115 vs ~1250; causal (4,12,4096) 215 vs 189, (16,12,2048) 262 vs 202,
  //  (4,12,8192) 661-701 vs 623-653 -> non-causal from 2048 keys on, causal never below 16384 rows)
  L.q64 = p->D == 64 && p->bias_mode != FAT5_BIAS_DENSE && !p->cu_seqlens_q && q64_env != 0 &&
          (q64_env == 1 || ((bh * ((p->M + 255) / 256) >= cu_scaled(512) || (bh * ((p->M + 255) / 256) >= cu_scaled(160) && p->N >= 8192)) &&  // ((4,12,1024x8192): 153 vs 166 us)
                            p->N >= ((p->bias_mode == FAT5_BIAS_RPE1D && bh * ((p->M + 255) / 256) < cu_scaled(1024)) ? 4096 : 2048) && (!p->causal || p->M >= 16384)));  // ((16,12,2048) T5 bias, 1536 workgroups: 309 vs 325 us)  // (T5 bias, band steps in the pipelined iteration since round 4:
                                                              //  (4,12,4096) 284-295 vs 298 us, (4,12,8192) 1089 vs 1107; (4,12,2048), 1.5 rounds: 97 vs 83 -> from 4096 keys on)
  // Both 64-wide bodies in ONE launch (attn_bwd_fused64_kernel; the dK/dV half forms its row statistics itself): one workgroup per CU
  // either way, so the two grids fill each other's empty last rounds -- and at cfg2 (96 + 96 workgroups) run side by side.
  const int f64_env = vsel(p->variant, FAT5_V_FUSED64_ON, FAT5_V_FUSED64_OFF);
  L.fused64 = false;
  if (p->D == 64 && p->bias_mode != FAT5_BIAS_DENSE && !p->cu_seqlens_q && f64_env != 0 &&
      smem_bwd_fused64_d64(p->rpe_radius, p->bias_mode) <= 160 * 1024) {
    // Measured, backward incl. the reduction launch, fused 64-wide against the library's previous choice (tools/attn_time.py, us):
    //   (4,12,512) T5 bias 30.4 vs 35.6, none 19.5 vs 23.2; (4,12,1024) 72.2 vs 76.9, 56.5 vs 60.0; (8,12,512) 49.2 vs 53.3, 35.4 vs 37.1;
    //   (2,12,512) 29.3 vs 40.8; (2,12,1024) 43.7 vs 51.8; (2,12,2048) 118.8 vs 128.6, 100.8 vs 111.2   -> up to 1.5 rounds of the chip;
    //   second sweep, fused vs not (T5 bias / none): (4,12,1280) 87.6 vs 102.1 / 76.5 vs 86.7; (4,12,1536) 136 vs 176 / 120 vs 150; (4,12,2048) 192.9-194.8 vs 196.0-199.4 / 171 vs 177;
    //   (4,12,2560) 291 vs 327 / 265 vs 289; (4,12,3072) 414 vs 424 / 376 vs 403; (4,12,3584) 548 vs 555 / 511 vs 501; (4,12,4096) 671 vs 650 / 625 vs 619;
    //   (16,12,512) 79.9 vs 89.1 / 62.3 vs 71.2; (16,12,1024) 223 vs 217 / 194 vs 189; (8,12,1024) 117 vs 125 / 98 vs 104; (2,12,4096) 351 vs 350; (1,12,8192) 670 vs 752;
    //   (1,12,4096) 214 vs 202 (the one miss inside the rule) -> up to 1152 workgroups (4.5 rounds) on roughly square problems;
    //   causal (4,12,512) 39.0 vs 33.7, (4,12,1024) 83.6 vs 61.7: the diagonal steps of the 64-wide bodies are unpipelined -> never.
    const long wq = bh * ((p->M + 255) / 256), tot = wg256 + wq;
    const long FUSED64_MAX_WG = cu_scaled(1152);
    // (a call that forces or forbids one of the 64-wide bodies / launch forms keeps that choice)
    const bool squarish = 2 * (p->M < p->N ? p->M : p->N) >= (p->M < p->N ? p->N : p->M);
    // causal with the T5 bias and the diagonal inside the band (-R <= N - M < R): the bias table in LDS carries the mask (-inf above the diagonal), the
    // diagonal steps run the pipelined band iteration (round 4).  Measured, one-launch 64-wide form vs the previous choice: (4,12,512) 29.7 vs 32.9 us,
    // (4,12,1024) 64.6 vs 60.8, (4,12,2048) 156.0 vs 174.5, (4,12,4096) 433.7 vs 428.1, (16,12,512) 76.5 vs 94.2, (16,12,1024) 208.0 vs 210.1
    const bool ctab = p->causal && p->bias_mode == FAT5_BIAS_RPE1D && p->N - p->M < p->rpe_radius && p->N - p->M >= -p->rpe_radius;
    // (round-4 audit: (2,12,2048) causal 80.9 vs 92.9 us -- the 384-workgroup exception holds at <= 1024 keys only; (8,12,2048) causal, 1536 workgroups: 269.8 vs 283.4)
    // (profiles/r05d_dispatch_audit_causal_bwd.log, longest-first order: causal (4,12,4096), 1536 workgroups, one launch 369.3 vs 387.8 us, T5 bias 382.6 vs 414.4;
    //  (2,32,4096), 2048 workgroups, T5 bias 506.1 vs 534.4, none 486.3 vs 493.5 -> causal problems up to 2048 workgroups)
    // (round-6 audit, profiles/r06_dispatch_audit_H12.log -- after causal problems left the mixed dK/dV launch the separate launches caught up at 1536 workgroups:
    //  (4,12,4096) causal 368.1 one launch vs 349.8 us, T5 bias 378.8 vs 377.2; (8,12,2048) 214.7 vs 206.1, T5 bias 226.8 vs 216.0; 768 workgroups stay:
    //  (4,12,2048) 115.4 / 119.1 one launch, (8,12,1024) 79.2 / 77.7; B H = 64 audit, 1024 workgroups: (2,32,2048) causal 146.1 one launch vs 139.3, T5 bias 152.3 vs 149.5,
    //  the same at (4,16,2048) / (8,8,2048) -> causal problems up to 896 workgroups)
    const long max_wg = (p->causal && (ctab || p->bias_mode == FAT5_BIAS_NONE)) ? cu_scaled(896) : FUSED64_MAX_WG;
    // (round 5, no bias: the dK/dV half's diagonal steps are pipelined (mask in the C operand) -- (16,12,512) causal 65.7 vs 72.1 us, (4,12,512) 21.7 vs 22.6;
    //  (4,12,1024) 45.9 either way -> up to 512 keys)
    // (closing audit of round 5, after causal launches went longest-first -- profiles/r05c_dispatch_audit_H12.log: T5 bias (4,12,1024), 384 workgroups, 44.6 one launch vs
    //  49.6 -> the 256 .. 512-workgroup exception holds below 1024 keys only; no bias (2,12,2048) 63.5 vs 70.3, (4,12,2048) 117.4 vs 127.7, (2,12,4096) 195.8 vs 198.7
    //  -> from 2048 keys on as well; (4,12,1024) 40.6 separate stays)
    const bool causal_ok = !p->causal || (ctab && squarish && (tot <= chip_cus() || tot >= cu_scaled(512) || p->N >= 1024)) ||
                           // ((3,5,2048) causal, 240 workgroups: 56.5 vs 70.9 us -- profiles/r05_dispatch_audit_H8_16_32.log; round-6 audit: (8,12,512) causal, 384 workgroups
                           //  -- one and a half rounds of unequal workgroups -- 36.9 one launch vs 28.5 us: at <= 512 keys one round or from two rounds on, as with the T5 bias)
                           // (B H = 64 audit: 1024 keys at exactly two rounds of workgroups -- (2,32,1024) / (4,16,1024) / (8,8,1024) causal 49.9-50.8 one launch vs 54.5-54.8 us; (8,12,1024), three rounds,
                           //  77.0 vs 79.2; at 384 / 480 workgroups the 32-wide one-launch form holds -> whole rounds of the chip, up to four)
                           (p->bias_mode == FAT5_BIAS_NONE && squarish && (tot <= chip_cus() || (p->N <= 512 && tot >= cu_scaled(512)) || p->N >= 2048 ||
                                                                           (p->N <= 1024 && tot % chip_cus() == 0 && tot <= 4L * chip_cus())));  // ((4,16,1536), 768 workgroups: 96.3 one launch vs 90.6 -> up to 1024 keys)
    // (off-grid audit of round 6, profiles/r06_audit_offgrid.log: the 64-wide bodies work in 256-key / 256-row workgroups -- 384 = 1.5 x 256 wastes a quarter of both halves:
    //  (6,12,384) 26.3 one launch vs 18.8 us on the 32-wide bodies, T5 bias 36.0 vs 27.3; (7,12,384) 37.4 vs 24.7 / 42.0 vs 31.6; (5,12,384) 26.1 vs 21.7 -> non-causal problems
    //  whose padding to 256 exceeds 20 % stay on the 32-wide bodies; causal (6,12,384) / (7,12,384) tie either way)
    const long padM = ((p->M + 255) / 256) * 256, padN = ((p->N + 255) / 256) * 256;
    //  (second pass, profiles/r06_audit_offgrid_after.log: only where the launch does not fit one round -- (4,12,384), 192 workgroups, 28.2 one launch vs 38.9; (5,12,384), 240: 28.4 vs
    //   30.5 -- and from 20 % on: (4,12,640), 288 workgroups, 38.9 one launch vs 26.9, T5 bias 53.1 vs 36.3; (5,12,640) T5 54.6 vs 46.3; (7,12,640) 56.9 vs 51.5; (6,12,640) ties)
    const bool padded = !p->causal && tot > chip_cus() && ((p->M > 256 && padM * 5 >= (long)p->M * 6) || (p->N > 256 && padN * 5 >= (long)p->N * 6));
    const bool rule = causal_ok && !padded && (tot <= cu_scaled(384) || (tot <= max_wg && squarish)) && b64_env != 0 && q64_env != 0 && kvh_env != 1 && mix_env != 1;
    L.fused64 = f64_env == 1 || rule;
  }
  if (L.fused64) {
    L.kv64 = L.q64 = true;
    L.kv64_half = false;
    L.kv64_mix_pf = -1;
    L.nw_kv = 4;
    L.n_nblk = (p->N + 255) / 256;
  }
  // Round 6: where the whole one-launch backward is resident at once (one workgroup per CU: cfg2's 96 + 96 on 256 CUs) the launch lasts as long as its longest
  // workgroup -- a dK/dV one, 57.4 k cycles against 41.0 k for dQ at cfg2 -- and ~8 k of those are the table gradient's per-diagonal sums: they move
  // to the dQ workgroups (attn_bwd_q64_body<..., QDG>; the dK/dV body is compiled without them).  The choice is part of the LAYOUT: a call that runs the stages
  // one by one launches the same two forms as separate kernels, so the partial rows always come from the same side.
  {
    const int qd_env = vsel(p->variant, FAT5_V_QDIAG_ON, FAT5_V_QDIAG_OFF);
    const long wq = bh * ((p->M + 255) / 256);
    L.diag_q = L.fused64 && p->bias_mode == FAT5_BIAS_RPE1D && (p->drpe1d || p->drpe_table) && qd_env != 0 &&
               smem_bwd_fused64_d64(p->rpe_radius, p->bias_mode) <= 160 * 1024 && (qd_env == 1 || wg256 + wq <= chip_cus());
  }
  // ... and both dense 64-wide bodies in ONE launch (attn_bwd_dfused64_kernel, round 5): the dense counterpart of fused64 -- the row statistics come from
  // bwd_stat2_kernel ahead of the launch instead of from the dK/dV half itself.  Separately the two launches of a short sequence leave most of the chip idle
  // twice ((4,12,512): 96 + 96 workgroups on 256 CUs) and those of a mid one each end in a half-empty round ((4,12,2048): 384 + 384).
  L.dfused64 = false;
  if (qdb_legal && dense_kv_ok && f64_env != 0 && qdb_env != 0 && b64_env != 0 && kvh_env != 1 && mix_env != 1 &&
      !(p->variant & (FAT5_V_DBIAS_STAGED | FAT5_V_DBIAS_INKERNEL)) && smem_bwd_kv64_d64(p->rpe_radius, p->bias_mode) <= 160 * 1024) {
    const long nq = (long)p->H * qdb_ngrp * ((p->M + 63) / 64), tot = wg256 + nq;
    // Measured, whole dense backward, one launch vs the same two bodies as separate launches (tools/attn_time.py --modes dense --what bwd, us; profiles/r05_dfused_time.log, r05_dfused_time2.log):
    //   both halves in one round of the chip: (4,12,512) 28.4 vs 45.8 (and 45.2 for the 32-wide launch + staged dS it replaces), causal 31.2 vs 50.4; (2,12,512) 26.7 vs 44.2 (47.8);
    //   (4,12,384) 31.5 vs 46.6; (8,12,256) 25.1 vs 36.6 (28.9); (4,12,256) 19.4 vs 29.8 (29.5); config 1's (2,8,128) 16.3 vs 23.1 (18.6)
    //   a round saved: (4,12,1536) 288 + 288 workgroups 164.6 vs 204.5; (6,12,1024) 288 + 384: 140 vs 165; (4,12,2048) 384 + 384: 230 vs 270, causal 182 vs 194; (8,12,1024) 150.5 vs 174.2;
    //   (16,12,512) 95.2 vs 104.2, causal 90.7 vs 101.0; (4,12,3072) 576 + 576: 550 vs 627
    //   no round saved -- the statistics kernel is a loss: (4,12,768) 144 + 144: 63.8 vs 60.5; (2,12,1024) 74.2 vs 71.9; (8,12,512) 192 + 192: 58.7 vs 55.1; (4,12,1024) 81.1 vs 77.6; (2,12,2048) 198 vs 189;
    //   (4,16,1024) 256 + 256: 88.5 vs 86.2; (16,12,256) 42.4 vs 39.6; (4,12,4096) 768 + 768: 925 vs 922; (16,12,1024) causal 272.6 vs 273.8; (8,12,2048) 630 vs 631.5
    // -> exactly when the one launch takes fewer rounds of the chip (one workgroup per CU either way) than the two launches together
    const long ncus = chip_cus();
    const long r_one = (tot + ncus - 1) / ncus, r_two = (wg256 + ncus - 1) / ncus + (nq + ncus - 1) / ncus;
    const bool rule = p->B >= 2 && r_one < r_two && r_two <= 8;
    if (f64_env == 1 || rule) {
      L.dfused64 = true;
      L.kv64 = true;
      L.kv64_half = false;
      L.kv64_mix_pf = -1;
      L.nw_kv = 4;
      L.n_nblk = (p->N + 255) / 256;
    }
  }
  if (L.q64) L.nw_q = 8;  // (256 query rows per workgroup)
  size_t off = 0;
  L.delta_off = off;
  // delta: (B,H,M) -- packed batches: (H, total_q), the layout of lse
  off = align_up(off + (p->cu_seqlens_q ? (size_t)p->H * p->total_q : (size_t)bh * p->M) * sizeof(float), 256);
  L.stat2_off = off;
  if (L.kv64) off = align_up(off + (size_t)bh * ((p->M + 31) / 32) * 64 * sizeof(float), 256);
  L.ds_staged = false;
  L.dbias_inkernel = false;
  L.ds_off = off;
  L.scratch_off = off;
  L.qdb64 = false;
  L.qdb_groups = 0;
  if (qdb_legal) {
    // Round 5: the reference's own operator -- one (1, H, M, N) bias for the whole batch (modeling_flash_t5.py:280-285) -- runs its dQ and the batch
    // sum of dS in ONE kernel (attn_bwd_qdb64.h): no (B, H, M, N) staging tensor, no third recomputation of S / dP.
    const int ngrp = qdb_ngrp;
    const bool legal = qdb_legal;
    // (a call that asks for one of the older dbias paths by variant bit keeps it)
    // (measured, whole backward, us, against the round-4 paths -- profiles/r05_dispatch_audit_none_dense_H12.log: (4,12,512) 49.4 vs 45.3, causal 55.1 vs 42.2: the one
    //  32-wide launch + staged dS stays ahead on the smallest problems; (4,12,1024) 81.8 vs 166.9, (16,12,512) 110.8 vs 118.9, (4,12,2048) 284 vs 436, (16,12,1024)
    //  causal 285 vs 330, (16,12,4096) 5014 vs 9162 -> from 2^25 scores per call on; second sweep, profiles/r05_dfused_time2.log: from 2^23 -- qdb_rule above)
    if (legal && qdb_env != 0 && (qdb_env == 1 || L.dfused64 || (qdb_rule && !(p->variant & (FAT5_V_DBIAS_STAGED | FAT5_V_DBIAS_INKERNEL))))) {
      L.qdb64 = true;
      L.qdb_groups = ngrp;
      L.q64 = false;
      L.fused64 = false;
      L.nw_q = 2;  // (64 query rows per workgroup: n_mblk = ceil(M / 64))
      if (ngrp > 1) off = align_up(off + (size_t)ngrp * p->H * p->M * p->N * sizeof(float), 256);
    }
  }
  if (p->bias_mode == FAT5_BIAS_DENSE && p->dbias && !L.qdb64) {
    const bool reduced = (p->dbias_batch != p->B) || (p->dbias_heads != p->H);
    // (variant FAT5_V_DBIAS_STAGED: the staged (B, H, M, N) + reduction path; FAT5_V_DBIAS_INKERNEL: the batch-inner kernel at any size)
    const int inker_env = (p->variant & FAT5_V_DBIAS_STAGED) ? 0 : ((p->variant & FAT5_V_DBIAS_INKERNEL) ? 2 : 1);
    // the model's case -- one bias per head shared by the batch -- reduces over the batch inside the dBias kernel: the
    // workspace stays O(B*H*M) (+ an fp32 (H, M, N) pass-through when the batch exceeds the kernel's 4-element register chunk)
    // (measured at (4,12,S,64): S = 512 staged 50 us vs 71 us; S = 2048 492 vs 461 us; S = 8192 6.26 vs 6.32 ms with a
    //  6.4 GB -> 1.6 MB workspace: the kernel takes over once the staging tensor would exceed 64 MB; =2 forces it)
    // (B > 4 costs the kernel one fp32 read-modify-write pass over (H, M, N) per 4 batch elements: at the reference's benchmark
    //  shape B = 16, S = 512 / 1024 the staged path is 1.8x / 1.35x faster (132 vs 235 us, 436 vs 590 us) -> staged up to 2 GB)
    const size_t staging = (size_t)bh * p->M * p->N * 2;
    // (head_dim 128, where this choice still decides the model's case -- at D = 64 the dQ + dBias body took it over --: the third recomputation costs twice the MFMA work
    //  of D = 64 while the staging tensor stays the same: (4,12,1024,128) staged 274 vs 341 us, (4,12,2048,128) 896 vs 933 (400 MB) -> staged up to 512 MB there;
    //  (16,12,1024,128) causal 713 vs 1077: profiles/r05_d128_bwd_variants.log)
    const bool big = p->B <= 4 ? staging > (size_t(p->D == 128 ? 512 : 64) << 20) : staging > (size_t(2) << 30);
    if (reduced && inker_env && (big || inker_env > 1) && p->dbias_batch == 1 && p->dbias_heads == p->H && p->bias_stride[0] == 0 &&
        (p->bias_stride[1] != 0 || p->H == 1) && p->unit_count == 0 && !p->cu_seqlens_q) {
      L.dbias_inkernel = true;
      if (p->B > 4) off = align_up(off + (size_t)p->H * p->M * p->N * sizeof(float), 256);
    } else if (reduced) {
      L.ds_staged = true;
      off = align_up(off + (size_t)bh * p->M * p->N * 2, 256);
    }
  }
  L.drpe_off = off;
  L.part_rows = L.diag_q ? (p->M + 255) / 256 : L.n_nblk;
  if (p->bias_mode == FAT5_BIAS_RPE1D && (p->drpe1d || p->drpe_table))
    off = align_up(off + (size_t)bh * L.part_rows * (2 * p->rpe_radius + 1) * sizeof(float), 256);
  L.total = off;
  return FAT5_OK;
}

// The layout is asked for two or three times per backward call (workspace size, stages, launches) and its mixed-launch model is a
// search: the last answer is kept per thread, keyed on every argument the function reads.
static int bwd_layout(const fat5_attn_params* p, BwdLayout& L) {
  uint32_t sm_bits;
  memcpy(&sm_bits, &p->sm_scale, 4);
  struct Key {
    int64_t v[24];
    bool operator==(const Key& o) const { return memcmp(v, o.v, sizeof(v)) == 0; }
  };
  const Key k = {{p->B, p->H, p->M, p->N, p->D, p->bias_mode, p->causal, (int64_t)p->variant, p->rpe_radius, p->dbias_batch, p->dbias_heads,
                  p->bias_stride[0], p->bias_stride[1], p->unit_count, p->total_q, p->cu_seqlens_q != nullptr, p->dbias != nullptr,
                  p->drpe1d != nullptr, p->drpe_table != nullptr, p->dtype, p->bias_stride[2], (int64_t)(reinterpret_cast<uintptr_t>(p->bias) & 15), (int64_t)sm_bits, 0}};
  thread_local Key last_k;
  thread_local BwdLayout last_L;
  thread_local int last_rc = -1;
  if (last_rc == FAT5_OK && last_k == k) {
    L = last_L;
    return FAT5_OK;
  }
  // (only successful answers are kept: a failing call runs again, so that fat5_last_error() is this call's message -- ADVICE r4.
  //  A rule in bwd_layout_compute that reads a NEW field of *p must add that field to Key above, or a stale layout is returned.)
  const int rc = bwd_layout_compute(p, L);
  last_k = k;
  last_L = L;
  last_rc = rc;
  return rc;
}

size_t fat5_attn_bwd_workspace_bytes(const fat5_attn_params* p) {
  if (check_common(p)) return 0;
  BwdLayout L;
  bwd_layout(p, L);
  return L.total;
}

static bool bwd_fusable(const BwdLayout& L, long grid_q, long grid_kv, int D, int variant) {
  if (D > 64) return false;  // (the D = 128 dK/dV body runs one wave per SIMD: no room for a co-resident dQ workgroup)
  const long fuse_max = 4L * chip_cus();  // measured: S=1024 (768 workgroups) +4 %, S=2048 (1536) -3 %
  return !(variant & FAT5_V_NO_FUSE) && !L.kv64 && !L.q64 && !L.qdb64 && L.nw_q == 4 && L.nw_kv == 4 && grid_q + grid_kv <= fuse_max;
}

int fat5_attn_bwd_launches(const fat5_attn_params* p) {
  if (check_common(p)) return 0;
  BwdLayout L;
  bwd_layout(p, L);
  const long bh = (long)p->B * p->H;
  const long grid_q = bh * ((p->M + 32 * L.nw_q - 1) / (32 * L.nw_q)), grid_kv = bh * L.n_nblk;
  return (bwd_fusable(L, grid_q, grid_kv, effD(p), p->variant) || L.fused64 || L.dfused64) ? 1 : 2;  // (dfused64: one launch of both bodies behind the small statistics kernel)
}

// Which kernel bodies a problem runs, as text (tests pin the dispatch rules with it; no device needed, no pointer of `p` is followed)
int fat5_attn_describe(const fat5_attn_params* p, char* out, size_t n) {
  int rc = check_common(p);
  if (rc) return rc;
  if (!out || n == 0) return fail(FAT5_EINVAL, "describe: no buffer");
  const FwdChoice fc = fwd_choice(p);
  BwdLayout L;
  bwd_layout(p, L);
  const long bh = (long)p->B * p->H;
  const long grid_q = bh * ((p->M + 32 * L.nw_q - 1) / (32 * L.nw_q)), grid_kv = bh * L.n_nblk;
  const bool fused = bwd_fusable(L, grid_q, grid_kv, effD(p), p->variant);
  char kv[48];
  if (L.kv64 && L.kv64_mix_pf > 0) snprintf(kv, sizeof kv, "64key-mixed:%d", L.kv64_mix_pf);
  else snprintf(kv, sizeof kv, "%s", L.kv64 ? (L.kv64_half ? "64key-half" : "64key") : "32key");
  snprintf(out, n, "fwd=%s dq=%s dkdv=%s fused=%d dbias=%s qdiag=%d", fc.fwd64 ? (fc.mixed ? "64row-mixed" : (fc.ksplit ? "64row-ksplit" : "64row")) : (fc.nw == -4 ? "32row-split" : "32row"),
           L.qdb64 ? "64row-batch4" : (L.q64 ? "64row" : "32row"), kv, (fused || L.fused64 || L.dfused64) ? 1 : 0,
           L.qdb64 ? (L.qdb_groups > 1 ? "dq-kernel+partials" : "dq-kernel") : (L.dbias_inkernel ? "inkernel" : (L.ds_staged ? "staged" : "direct")),
           L.diag_q ? 1 : 0);  // (qdiag: T5 bias, the table gradient's per-diagonal sums come from the dQ workgroups)
  return FAT5_OK;
}

int fat5_attn_bwd(const fat5_attn_params* p, void* stream_) { return fat5_attn_bwd_stages(p, FAT5_BWD_ALL, stream_); }

int fat5_attn_bwd_stages(const fat5_attn_params* p, int stages, void* stream_) {
  int rc = check_common(p);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  if ((p->cu_seqlens_q == nullptr) != (p->cu_seqlens_k == nullptr)) return fail(FAT5_EINVAL, "cu_seqlens_q/k must both be set");
  if (p->cu_seqlens_q && p->bias_mode == FAT5_BIAS_DENSE) return fail(FAT5_EINVAL, "varlen: dense bias unsupported (none or rpe1d)");
  if (!p->q || !p->k || !p->v || !p->o || !p->lse || !p->dout || !p->dq || !p->dk || !p->dv)
    return fail(FAT5_EINVAL, "bwd: null tensor pointer");
  if (!strides_ok(p->q, p->q_stride) || !strides_ok(p->k, p->k_stride) || !strides_ok(p->v, p->v_stride) ||
      !strides_ok(p->o, p->o_stride) || !strides_ok(p->dout, p->do_stride) || !strides_ok(p->dq, p->dq_stride) ||
      !strides_ok(p->dk, p->dk_stride) || !strides_ok(p->dv, p->dv_stride))
    return fail(FAT5_EINVAL, "bwd: tensors must be 16-byte aligned with strides that are multiples of 8 elements");
  if (p->bias_mode == FAT5_BIAS_DENSE && p->dbias) {
    if (!((p->dbias_batch == 1 || p->dbias_batch == p->B) && (p->dbias_heads == 1 || p->dbias_heads == p->H)))
      return fail(FAT5_EINVAL, "bwd: dbias batch/heads (%d,%d) must be 1 or (B,H)", p->dbias_batch, p->dbias_heads);
    if (!aligned16(p->dbias)) return fail(FAT5_EINVAL, "bwd: dbias must be 16-byte aligned");
    if (p->unit_count > 0 && (p->dbias_batch != p->B || p->dbias_heads != p->H))
      return fail(FAT5_EINVAL, "bwd: a unit range writes dense dbias only in its unreduced (B, H, M, N) form");
  }
  // the kernels address one (b,h) slice of every tensor through a 32-bit buffer descriptor (like the forward)
  if (!slice_fits(p->M, p->q_stride[2], p->D) || !slice_fits(p->N, p->k_stride[2], p->D) || !slice_fits(p->N, p->v_stride[2], p->D) ||
      !slice_fits(p->M, p->o_stride[2], p->D) || !slice_fits(p->M, p->do_stride[2], p->D) || !slice_fits(p->M, p->dq_stride[2], p->D) ||
      !slice_fits(p->N, p->dk_stride[2], p->D) || !slice_fits(p->N, p->dv_stride[2], p->D))
    return fail(FAT5_EINVAL, "bwd: one (batch, head) slice must span less than 2 GiB");
  BwdLayout L;
  bwd_layout(p, L);
  if (p->bias_mode == FAT5_BIAS_RPE1D) {
    // the dK/dV body keeps the table and one private diagonal accumulator per wave in LDS
    const size_t lds = L.kv64 ? (L.kv64_half ? smem_bwd_kv64h_d64(p->rpe_radius, p->bias_mode) : smem_bwd_kv64_d64(p->rpe_radius, p->bias_mode))
                     : effD(p) == 32 ? smem_bwd_kv_d32(L.nw_kv, p->rpe_radius, p->bias_mode)
                                  : (p->D == 64 ? smem_bwd_kv_d64(L.nw_kv, p->rpe_radius, p->bias_mode) : smem_bwd_kv_d128(L.nw_kv, p->rpe_radius, p->bias_mode));
    if (lds > 160 * 1024)
      return fail(FAT5_EINVAL, "bwd: rpe_radius %d needs %zu bytes of LDS (160 KiB per workgroup; radius <= 1024 fits every head_dim)", p->rpe_radius, lds);
  }
  if (L.total > 0 && (!p->workspace || p->workspace_bytes < L.total))
    return fail(FAT5_EWORKSPACE, "bwd: workspace of %zu bytes required, got %zu", L.total, p->workspace_bytes);
  if ((reinterpret_cast<uintptr_t>(p->workspace) & 255) != 0) return fail(FAT5_EINVAL, "bwd: workspace must be 256-byte aligned");
  char* ws = (char*)p->workspace;

  AttnArgs a;
  fill_common(p, a);
  a.dout = (const uint16_t*)p->dout; a.dq = (uint16_t*)p->dq; a.dk = (uint16_t*)p->dk; a.dv = (uint16_t*)p->dv;
  for (int i = 0; i < 3; ++i) {
    a.dos[i] = p->do_stride[i]; a.dqs[i] = p->dq_stride[i]; a.dks[i] = p->dk_stride[i]; a.dvs[i] = p->dv_stride[i];
  }
  a.delta = (float*)(ws + L.delta_off);
  a.stat2 = L.kv64 ? (float*)(ws + L.stat2_off) : nullptr;
  // the dK/dV kernel stages -L/scale (accumulator initial value): an exact zero scale (softmax of the bias alone,
  // dq = dk = 0) runs with 1e-30 -- q.k * 1e-30 and the 1e-30-scaled dq / dk vanish in fp32 / the output dtype
  if (a.scale == 0.f) a.scale = 1e-30f;
  const int64_t MN = (int64_t)p->M * p->N;
  const long bh = (long)p->B * p->H;
  const bool bf16 = p->dtype == FAT5_BF16;
  if (p->bias_mode == FAT5_BIAS_DENSE && p->dbias && L.qdb64) {
    // (tiles above the causal diagonal are never visited: zeros by definition, reference :153,:160 -- the kernel writes them itself, row block by row
    //  block; with partial slabs the reduction knows the mask.  No memset.)
  } else if (p->bias_mode == FAT5_BIAS_DENSE && p->dbias && !L.dbias_inkernel) {
    if (L.ds_staged) {
      a.ds_out = (uint16_t*)(ws + L.ds_off);
    } else {
      a.ds_out = (uint16_t*)p->dbias;
    }
    a.dss[0] = (int64_t)p->H * MN; a.dss[1] = MN; a.dss[2] = p->N;
    a.ds_vec4 = (p->N % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.ds_out) & 7) == 0);
    a.ds_vec8 = (p->N % 8 == 0) && ((reinterpret_cast<uintptr_t>(a.ds_out) & 15) == 0);
    // tiles above the diagonal are never visited (reference zero-fills too, :153,:160). Staged dS: the reduction knows the mask and
    // neither reads nor needs them (no 2-byte-per-score memset, half the reduction's reads); dS written straight into dbias: zero-fill
    if (p->causal && (stages & FAT5_BWD_DQ) && !L.ds_staged) {
      // (unit u = h * B + b occupies row b * H + h of the (B, H, M, N) tensor: a range is not contiguous there -> per unit)
      if (p->unit_count > 0) {
        for (int u = p->unit_begin; u < p->unit_begin + p->unit_count; ++u) {
          const int hh = u / p->B, bb = u - hh * p->B;
          hipError_t e = hipMemsetAsync(a.ds_out + ((size_t)bb * p->H + hh) * MN, 0, (size_t)MN * 2, stream);
          if (e != hipSuccess) return hip_fail(e, "memset ds");
        }
      } else {
        hipError_t e = hipMemsetAsync(a.ds_out, 0, (size_t)bh * MN * 2, stream);
        if (e != hipSuccess) return hip_fail(e, "memset ds");
      }
    }
  }
  if (p->bias_mode == FAT5_BIAS_RPE1D && (p->drpe1d || p->drpe_table)) {
    if (p->drpe_table && (!p->rpe_bucket || p->rpe_num_buckets <= 0))
      return fail(FAT5_EINVAL, "bwd: drpe_table needs rpe_bucket and rpe_num_buckets");
    a.drpe_part = (float*)(ws + L.drpe_off);
  }

  a.n_mblk = (p->M + 32 * L.nw_q - 1) / (32 * L.nw_q);
  a.n_nblk = L.n_nblk;
  a.n_kv_blocks = 0;
  a.diag_q = L.diag_q ? 1 : 0;
  a.part_stride = L.part_rows;
  const long grid_q = n_units(p) * a.n_mblk, grid_kv = n_units(p) * a.n_nblk;
  const long full_q = bh * a.n_mblk, full_kv = bh * a.n_nblk;  // (variant choice: see fat5_attn_fwd)
  // Short sequences: both grids together fit the chip at two workgroups per CU -> one launch, the two halves run
  // side by side (attn_bwd_fused_kernel).  fat5_attn_params.variant & FAT5_V_NO_FUSE forbids it (tests / profiling: _lib.variant()).
  const bool fuse = (stages & FAT5_BWD_DQ) && (stages & FAT5_BWD_DKDV) && bwd_fusable(L, full_q, full_kv, effD(p), p->variant);
  if (fuse) {
    a.n_kv_blocks = (int)grid_kv;
    launch_fn fn = effD(p) == 32 ? launch_bwd_fused_d32 : (p->D == 64 ? launch_bwd_fused_d64 : launch_bwd_fused_d128);
    hipError_t e = fn(a, bf16, p->bias_mode, 4, (int)(grid_q + grid_kv), stream);
    if (e != hipSuccess) return hip_fail(e, "attn_bwd_fused launch");
  } else if (L.fused64 && (stages & FAT5_BWD_DQ) && (stages & FAT5_BWD_DKDV)) {
    // (a call for one stage only runs the same two bodies as separate launches, the row statistics through the workspace)
    a.n_kv_blocks = (int)grid_kv;
    a.stat2 = nullptr;
    hipError_t e = launch_bwd_fused64_d64(a, bf16, p->bias_mode, 4, (int)(grid_q + grid_kv), stream);
    if (e != hipSuccess) return hip_fail(e, "attn_bwd_fused64 launch");
  } else if (L.dfused64 && (stages & FAT5_BWD_DQ) && (stages & FAT5_BWD_DKDV)) {
    // dense bias: row statistics, then the dense dK/dV workgroups and the dQ + dBias ones side by side (a call for one stage runs the same bodies as separate launches)
    a.n_kv_blocks = (int)grid_kv;
    a.part_stride = a.n_nblk;
    const long g = (long)p->H * L.qdb_groups * ((p->M + 63) / 64);
    if (g + grid_kv + 16 > 0x7fffffffL) return fail(FAT5_EINVAL, "grid too large");
    hipError_t e = launch_bwd_dfused64_d64(a, bf16, L.qdb_groups > 1 ? (void*)(ws + L.scratch_off) : p->dbias, L.qdb_groups > 1, (int)g, stream);
    if (e != hipSuccess) return hip_fail(e, "attn_bwd_dfused64 launch");
  } else {
    // 1) dQ (+ delta)
    if ((stages & FAT5_BWD_DQ) && L.qdb64) {
      // ... and the batch-reduced dbias (or its fp32 slabs) in the same launch
      const long g = (long)p->H * L.qdb_groups * ((p->M + 63) / 64);
      if (g > 0x7fffffffL) return fail(FAT5_EINVAL, "grid too large");
      hipError_t e = launch_bwd_qdb64_d64(a, bf16, L.qdb_groups > 1 ? (void*)(ws + L.scratch_off) : p->dbias, L.qdb_groups > 1, (int)g, stream);
      if (e != hipSuccess) return hip_fail(e, "attn_bwd_qdb64 launch");
    } else if (stages & FAT5_BWD_DQ) {
      launch_fn fn = effD(p) == 32 ? launch_bwd_q_d32 : (p->D == 64 ? launch_bwd_q_d64 : launch_bwd_q_d128);
      if (L.q64) fn = launch_bwd_q64_d64;
      hipError_t e = fn(a, bf16, p->bias_mode, L.nw_q, (int)grid_q, stream);
      if (e != hipSuccess) return hip_fail(e, "attn_bwd_q launch");
    }
    // 2) dK, dV, dBias
    if (stages & FAT5_BWD_DKDV) {
      launch_fn fn = effD(p) == 32 ? launch_bwd_kv_d32 : (p->D == 64 ? launch_bwd_kv_d64 : launch_bwd_kv_d128);
      if (L.kv64) fn = launch_bwd_kv64_d64;
      hipError_t e = hipSuccess;
      if (L.kv64 && L.kv64_mix_pf > 0 && p->unit_count == 0) {
        // (a.n_nblk counts 128-key rows; the 256-key workgroups cover two of them each)
        a.mix_full = L.kv64_mix_pf;
        const long gkv = 8L * ((long)L.kv64_mix_pf * ((p->N + 255) / 256) + (bh / 8 - L.kv64_mix_pf) * (long)a.n_nblk);
        e = fn(a, bf16, p->bias_mode, 3, (int)gkv, stream);
      } else if (L.kv64 && L.kv64_mix_pf > 0) {
        // a unit range of a problem whose whole-problem launch is mixed: units [0, 8 pf) are the 256-key pairs (head-major numbering,
        // attn_bwd_kv64_mixed_kernel) -> at most one 256-key and one half-length launch, every pair through the body it has unsharded
        const int split = 8 * L.kv64_mix_pf, ub = p->unit_begin, ue = p->unit_begin + p->unit_count;
        AttnArgs af = a;
        if (ub < std::min(ue, split)) {
          af.unit_begin = ub; af.unit_count = std::min(ue, split) - ub;
          af.n_nblk = (p->N + 255) / 256; af.part_rows2 = 1;
          e = fn(af, bf16, p->bias_mode, 4, (int)((long)af.unit_count * af.n_nblk), stream);
        }
        if (e == hipSuccess && std::max(ub, split) < ue) {
          af = a;
          af.unit_begin = std::max(ub, split); af.unit_count = ue - af.unit_begin;
          e = fn(af, bf16, p->bias_mode, 2, (int)((long)af.unit_count * af.n_nblk), stream);
        }
      } else {
        e = fn(a, bf16, p->bias_mode, L.nw_kv, (int)grid_kv, stream);
      }
      if (e != hipSuccess) return hip_fail(e, "attn_bwd_kv launch");
    }
  }
  // 3) bias gradient: batch-inner kernel (reads delta: after the dQ stage), or reductions over the broadcast dims
  if (L.dbias_inkernel && (stages & FAT5_BWD_REDUCE)) {
    AttnArgs ab = a;
    ab.n_mblk = (p->M + 127) / 128;
    ab.batch_inner = 0;
    ab.lds_stage = (p->D <= 64 && !(p->variant & FAT5_V_DBIAS_NOSPLIT)) ? 1 : 0;  // (two wave groups sharing the batch: two waves per SIMD)
    // key tiles of a strip are independent: split them over workgroups until the grid covers the chip about twice
    const long strips = (long)p->H * ab.n_mblk, ntile = (p->N + 63) / 64;
    long nsplit = 1;
    while (strips * nsplit < cu_scaled(512) && nsplit * 2 <= ntile) nsplit *= 2;
    ab.n_nblk = (int)nsplit;
    const long grid = strips * nsplit;
    typedef hipError_t (*dbias_fn)(const AttnArgs&, int, void*, float*, int, hipStream_t);
    dbias_fn fn = effD(p) == 32 ? launch_bwd_dbias_d32 : (p->D == 64 ? launch_bwd_dbias_d64 : launch_bwd_dbias_d128);
    hipError_t e = fn(ab, bf16, p->dbias, p->B > 4 ? (float*)(ws + L.scratch_off) : nullptr, (int)grid, stream);
    if (e != hipSuccess) return hip_fail(e, "attn_bwd_dbias launch");
  }
  if (L.qdb64 && L.qdb_groups > 1 && (stages & FAT5_BWD_REDUCE)) {
    hipError_t e = launch_dbias_partial_reduce((const float*)(ws + L.scratch_off), p->dbias, bf16, L.qdb_groups, p->H, p->M, p->N, p->causal, stream);
    if (e != hipSuccess) return hip_fail(e, "dbias_partial_reduce launch");
  }
  if (L.ds_staged && (stages & FAT5_BWD_REDUCE)) {
    const int64_t chunks = (MN + 7) / 8 * p->dbias_batch * p->dbias_heads;
    const int grid = (int)((chunks + 255) / 256);
    if (bf16)
      hipLaunchKernelGGL(dbias_reduce_kernel<true>, dim3(grid), dim3(256), 0, stream, a.ds_out, (uint16_t*)p->dbias, p->B,
                         p->H, p->dbias_batch, p->dbias_heads, MN, p->causal ? p->N : 0, p->N - p->M);
    else
      hipLaunchKernelGGL(dbias_reduce_kernel<false>, dim3(grid), dim3(256), 0, stream, a.ds_out, (uint16_t*)p->dbias, p->B,
                         p->H, p->dbias_batch, p->dbias_heads, MN, p->causal ? p->N : 0, p->N - p->M);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "dbias_reduce launch");
  }
  if (a.drpe_part && (stages & FAT5_BWD_REDUCE)) {
    const int n1 = 2 * p->rpe_radius + 1;
    const size_t smem = (size_t)n1 * 24;
    if (smem > 48 * 1024) {
      hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(drpe_reduce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (ea != hipSuccess) return hip_fail(ea, "drpe_reduce attribute");
    }
    hipLaunchKernelGGL(drpe_reduce_kernel, dim3(p->H), dim3(1024), smem, stream, a.drpe_part, p->drpe1d, p->rpe_bucket,
                       p->drpe_table, p->B, p->H, L.part_rows, n1, p->rpe_num_buckets, p->unit_begin, p->unit_count, div_magic(n1, 4L * n1),
                       div_magic(L.part_rows, (long)p->B * L.part_rows));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "drpe_reduce launch");
  }
  return FAT5_OK;
}

int fat5_rpe1d_from_table(const void* table, int table_dtype, const int32_t* rpe_bucket, float* rpe1d, int32_t H, int32_t rpe_radius,
                          int32_t num_buckets, void* stream_) {
  if (!table || !rpe_bucket || !rpe1d) return fail(FAT5_EINVAL, "rpe1d_from_table: null pointer");
  if (table_dtype != FAT5_F32 && table_dtype != FAT5_F16 && table_dtype != FAT5_BF16) return fail(FAT5_EINVAL, "rpe1d_from_table: bad dtype");
  if (H <= 0 || num_buckets <= 0 || rpe_radius < 1 || rpe_radius > 2048) return fail(FAT5_EINVAL, "rpe1d_from_table: bad shape");
  const int n1 = 2 * rpe_radius + 1;
  const int grid = (H * n1 + 255) / 256;
  hipStream_t stream = (hipStream_t)stream_;
  dispatch_dtype(table_dtype, [&](auto dt_) {
    constexpr int DT = decltype(dt_)::value;
    hipLaunchKernelGGL(rpe1d_gather_kernel<DT>, dim3(grid), dim3(256), 0, stream, table, rpe_bucket, rpe1d, H, n1, num_buckets);
  });
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rpe1d_gather launch");
  return FAT5_OK;
}

// ============================================================================================
// RMSNorm
// ============================================================================================
#define RMS_DISPATCH(...)                                  \
  dispatch_dtype(x_dtype, [&](auto xt_) {                  \
    dispatch_dtype(w_dtype, [&](auto wt_) {                \
      constexpr int XDT = decltype(xt_)::value;            \
      constexpr int WDT = decltype(wt_)::value;            \
      __VA_ARGS__                                          \
    });                                                    \
  });

static int dtype_ok(int d) { return d == FAT5_F32 || d == FAT5_F16 || d == FAT5_BF16; }
static int vec_of(int d) { return d == FAT5_F32 ? 4 : 8; }

int fat5_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int64_t n, int64_t xs,
                     int64_t ys, float eps, int x_dtype, int w_dtype, void* stream_) {
  if (!x || !w || !y || !rstd) return fail(FAT5_EINVAL, "rmsnorm_fwd: null pointer");
  if (!dtype_ok(x_dtype) || !dtype_ok(w_dtype)) return fail(FAT5_EINVAL, "rmsnorm_fwd: bad dtype");
  if (rows <= 0 || n <= 0 || n > (1 << 24)) return fail(FAT5_EINVAL, "rmsnorm_fwd: bad shape");
  hipStream_t stream = (hipStream_t)stream_;
  const int vx = vec_of(x_dtype);
  const bool vecok = (n % vx == 0) && (xs % vx == 0) && (ys % vx == 0) && aligned16(x) && aligned16(y) && aligned16(w) &&
                     (n % 8 == 0);
  const int grid = (int)((rows + 3) / 4);
  RMS_DISPATCH({
    if (vecok)
      hipLaunchKernelGGL((rmsnorm_fwd_kernel<XDT, WDT, true>), dim3(grid), dim3(256), 0, stream, x, w, y, rstd, rows, (int)n, xs, ys, eps);
    else
      hipLaunchKernelGGL((rmsnorm_fwd_kernel<XDT, WDT, false>), dim3(grid), dim3(256), 0, stream, x, w, y, rstd, rows, (int)n, xs, ys, eps);
  })
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rmsnorm_fwd launch");
  return FAT5_OK;
}

int fat5_add_rmsnorm_fwd(const void* x, const void* r, const void* w, void* h, void* y, float* rstd, int64_t rows, int64_t n,
                         int64_t xs, int64_t rs, int64_t hs, int64_t ys, float eps, int x_dtype, int w_dtype, void* stream_) {
  if (!x || !r || !w || !h || !y || !rstd) return fail(FAT5_EINVAL, "add_rmsnorm_fwd: null pointer");
  if (!dtype_ok(x_dtype) || !dtype_ok(w_dtype)) return fail(FAT5_EINVAL, "add_rmsnorm_fwd: bad dtype");
  if (rows <= 0 || n <= 0 || n > (1 << 24)) return fail(FAT5_EINVAL, "add_rmsnorm_fwd: bad shape");
  hipStream_t stream = (hipStream_t)stream_;
  const int vx = vec_of(x_dtype);
  const bool vecok = (n % vx == 0) && (xs % vx == 0) && (rs % vx == 0) && (hs % vx == 0) && (ys % vx == 0) && aligned16(x) &&
                     aligned16(r) && aligned16(h) && aligned16(y) && aligned16(w) && (n % 8 == 0);
  const int grid = (int)((rows + 3) / 4);
  const int nch = (int)((n + 64 * vx - 1) / (64 * vx));
  RMS_DISPATCH({
    if (vecok && nch <= 2)
      hipLaunchKernelGGL((add_rmsnorm_fwd_reg_kernel<XDT, WDT, 2>), dim3(grid), dim3(256), 0, stream, x, r, w, h, y, rstd, rows, (int)n, xs, rs, hs, ys, eps);
    else if (vecok && nch <= 4)
      hipLaunchKernelGGL((add_rmsnorm_fwd_reg_kernel<XDT, WDT, 4>), dim3(grid), dim3(256), 0, stream, x, r, w, h, y, rstd, rows, (int)n, xs, rs, hs, ys, eps);
    else if (vecok)
      hipLaunchKernelGGL((add_rmsnorm_fwd_kernel<XDT, WDT, true>), dim3(grid), dim3(256), 0, stream, x, r, w, h, y, rstd, rows, (int)n, xs, rs, hs, ys, eps);
    else
      hipLaunchKernelGGL((add_rmsnorm_fwd_kernel<XDT, WDT, false>), dim3(grid), dim3(256), 0, stream, x, r, w, h, y, rstd, rows, (int)n, xs, rs, hs, ys, eps);
  })
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "add_rmsnorm_fwd launch");
  return FAT5_OK;
}

static int rms_bwd_blocks(int64_t rows) {
  // persistent 8-wave workgroups, one row per wave and trip: two per CU keep enough 16-byte
  // loads in flight to cover the HBM latency (one per CU: 4.5 TB/s at (65536, 1024))
  // (measured: 256 / 384 / 512 / 768 workgroups -> 4.48 / 4.98 / 5.08 / 4.78 TB/s; small inputs keep >= 2 rows per wave so that
  //  the dw reduction over the workgroups' partial sums stays short)
  const int cap = 512;
  int64_t b = (rows + 15) / 16;
  return (int)(b < cap ? b : cap);
}
size_t fat5_rmsnorm_bwd_workspace_bytes(int64_t rows, int64_t n) {
  return (size_t)rms_bwd_blocks(rows) * (size_t)n * sizeof(float);
}

static int rmsnorm_bwd_impl(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, int64_t drs, void* dx,
                            void* dw, int64_t rows, int64_t n, int64_t dys, int64_t xs, int64_t dxs, int x_dtype, int w_dtype,
                            void* workspace, size_t workspace_bytes, void* stream_);
int fat5_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, void* dx, void* dw, int64_t rows,
                     int64_t n, int64_t dys, int64_t xs, int64_t dxs, int x_dtype, int w_dtype, void* workspace,
                     size_t workspace_bytes, void* stream_) {
  return rmsnorm_bwd_impl(dy, x, w, rstd, nullptr, 0, dx, dw, rows, n, dys, xs, dxs, x_dtype, w_dtype, workspace, workspace_bytes, stream_);
}
int fat5_add_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dres, void* dx, void* dw,
                         int64_t rows, int64_t n, int64_t dys, int64_t hs, int64_t drs, int64_t dxs, int x_dtype, int w_dtype,
                         void* workspace, size_t workspace_bytes, void* stream_) {
  return rmsnorm_bwd_impl(dy, h, w, rstd, dres, drs, dx, dw, rows, n, dys, hs, dxs, x_dtype, w_dtype, workspace, workspace_bytes, stream_);
}
static int rmsnorm_bwd_impl(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, int64_t drs, void* dx,
                            void* dw, int64_t rows, int64_t n, int64_t dys, int64_t xs, int64_t dxs, int x_dtype, int w_dtype,
                            void* workspace, size_t workspace_bytes, void* stream_) {
  if (!dy || !x || !w || !rstd || !dx || !dw) return fail(FAT5_EINVAL, "rmsnorm_bwd: null pointer");
  if (!dtype_ok(x_dtype) || !dtype_ok(w_dtype)) return fail(FAT5_EINVAL, "rmsnorm_bwd: bad dtype");
  if (rows <= 0 || n <= 0) return fail(FAT5_EINVAL, "rmsnorm_bwd: bad shape");
  const int vx = vec_of(x_dtype);
  const bool vecok = !((n % 8) || (xs % vx) || (dys % vx) || (dxs % vx) || !aligned16(x) || !aligned16(dy) || !aligned16(dx) ||
                       !aligned16(w) || (dres && ((drs % vx) || !aligned16(dres)))) && (n <= 16 * 64 * vx);
  if (n > 16384) return fail(FAT5_EINVAL, "rmsnorm_bwd: n = %lld exceeds 16384", (long long)n);
  const size_t need = fat5_rmsnorm_bwd_workspace_bytes(rows, n);
  if (!workspace || workspace_bytes < need) return fail(FAT5_EWORKSPACE, "rmsnorm_bwd: workspace of %zu bytes required", need);
  hipStream_t stream = (hipStream_t)stream_;
  const int blocks = rms_bwd_blocks(rows);
  const int nch = (int)((n + 64 * vx - 1) / (64 * vx));
  const size_t smem = (size_t)n * sizeof(float);
  float* part = (float*)workspace;
#define RMS_BWD_LAUNCH(NCH) \
  hipLaunchKernelGGL((rmsnorm_bwd_kernel<XDT, WDT, NCH>), dim3(blocks), dim3(512), smem, stream, dy, x, w, rstd, dx, part, rows, (int)n, dys, xs, dxs, dres, drs)
  RMS_DISPATCH({
    if (!vecok) {
      auto kern = rmsnorm_bwd_scalar_kernel<XDT, WDT>;
      if (smem > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem, stream, dy, x, w, rstd, dx, part, rows, (int)n, dys, xs, dxs, dres, drs);
    } else if (nch <= 2) RMS_BWD_LAUNCH(2);
    else if (nch <= 4) RMS_BWD_LAUNCH(4);
    else if (nch <= 8) RMS_BWD_LAUNCH(8);
    else RMS_BWD_LAUNCH(16);
  })
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rmsnorm_bwd launch");
  const int g2 = (int)((n + 63) / 64);
  switch (w_dtype) {
    case FAT5_F32: hipLaunchKernelGGL(rmsnorm_dw_reduce_kernel<FAT5_F32>, dim3(g2), dim3(1024), 0, stream, part, dw, blocks, (int)n); break;
    case FAT5_F16: hipLaunchKernelGGL(rmsnorm_dw_reduce_kernel<FAT5_F16>, dim3(g2), dim3(1024), 0, stream, part, dw, blocks, (int)n); break;
    default: hipLaunchKernelGGL(rmsnorm_dw_reduce_kernel<FAT5_BF16>, dim3(g2), dim3(1024), 0, stream, part, dw, blocks, (int)n); break;
  }
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rmsnorm_dw_reduce launch");
  return FAT5_OK;
}

// ============================================================================================
// Stacked projection weights (fold_weights.h)
// ============================================================================================
int fat5_fold_weights(const void* w0, const void* w1, const void* w2, int64_t n0, int64_t n1, int64_t n2, int64_t ld0, int64_t ld1,
                      int64_t ld2, const void* g, void* out, int64_t K, int dtype, void* stream_) {
  if (!w0 || !out || n0 <= 0 || n1 < 0 || n2 < 0 || (n1 > 0 && !w1) || (n2 > 0 && !w2)) return fail(FAT5_EINVAL, "fold_weights: bad arguments");
  if (dtype != FAT5_F16 && dtype != FAT5_BF16) return fail(FAT5_EINVAL, "fold_weights: 16-bit dtypes only");
  if (K <= 0 || K % 8 != 0 || ld0 % 8 || ld1 % 8 || ld2 % 8 || !aligned16(w0) || !aligned16(out) || (w1 && !aligned16(w1)) ||
      (w2 && !aligned16(w2)) || (g && !aligned16(g)))
    return fail(FAT5_EINVAL, "fold_weights: K and the row strides must be multiples of 8 elements, bases 16-byte aligned");
  const int64_t items = (n0 + n1 + n2) * (K / 8);
  if (n0 + n1 + n2 > 0x7fffffffLL || items > (int64_t)0x7fffffff * 256) return fail(FAT5_EINVAL, "fold_weights: too large");
  hipStream_t stream = (hipStream_t)stream_;
  const unsigned grid = (unsigned)((items + 255) / 256);
  if (dtype == FAT5_BF16)
    hipLaunchKernelGGL(fold_weights_kernel<true>, dim3(grid), dim3(256), 0, stream, (const uint16_t*)w0, (const uint16_t*)w1, (const uint16_t*)w2,
                       (int)n0, (int)n1, (int)n2, ld0, ld1, ld2, (const uint16_t*)g, (uint16_t*)out, (int)K);
  else
    hipLaunchKernelGGL(fold_weights_kernel<false>, dim3(grid), dim3(256), 0, stream, (const uint16_t*)w0, (const uint16_t*)w1, (const uint16_t*)w2,
                       (int)n0, (int)n1, (int)n2, ld0, ld1, ld2, (const uint16_t*)g, (uint16_t*)out, (int)K);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "fold_weights launch");
  return FAT5_OK;
}

// rows of one slab of fold_weights_bwd_kernel: at most 64 slabs, at least 64 rows each (a multiple of the 32 row phases)
static int fold_bwd_rows_per(int64_t ntot) { return (int)std::max<int64_t>(64, ((ntot + 63) / 64 + 31) / 32 * 32); }
size_t fat5_fold_weights_bwd_scratch_bytes(int64_t n_total, int64_t K) {
  if (n_total <= 0 || K <= 0) return 0;
  const int rp = fold_bwd_rows_per(n_total);
  return (size_t)((n_total + rp - 1) / rp) * (size_t)K * sizeof(float);
}

int fat5_fold_weights_bwd(const void* dwg, const void* w0, const void* w1, const void* w2, int64_t n0, int64_t n1, int64_t n2, int64_t ld0,
                          int64_t ld1, int64_t ld2, const void* g, void* dw0, void* dw1, void* dw2, void* dg, int64_t K, int dtype,
                          void* scratch, size_t scratch_bytes, void* stream_) {
  if (!dwg || !w0 || !g || n0 <= 0 || n1 < 0 || n2 < 0 || (n1 > 0 && !w1) || (n2 > 0 && !w2)) return fail(FAT5_EINVAL, "fold_weights_bwd: bad arguments");
  if (dtype != FAT5_F16 && dtype != FAT5_BF16) return fail(FAT5_EINVAL, "fold_weights_bwd: 16-bit dtypes only");
  if (K <= 0 || K % 64 != 0 || ld0 % 8 || ld1 % 8 || ld2 % 8) return fail(FAT5_EINVAL, "fold_weights_bwd: K must be a multiple of 64, row strides of 8");
  const void* ptrs[] = {dwg, w0, w1, w2, g, dw0, dw1, dw2};
  for (const void* q : ptrs)
    if (q && !aligned16(q)) return fail(FAT5_EINVAL, "fold_weights_bwd: 16-byte aligned bases");
  const int64_t ntot = n0 + n1 + n2;
  if (ntot > 0x7fffffffLL) return fail(FAT5_EINVAL, "fold_weights_bwd: too large");
  if (dg && (!scratch || scratch_bytes < fat5_fold_weights_bwd_scratch_bytes(ntot, K)))
    return fail(FAT5_EWORKSPACE, "fold_weights_bwd: dg needs %zu bytes of scratch (fat5_fold_weights_bwd_scratch_bytes)",
                fat5_fold_weights_bwd_scratch_bytes(ntot, K));
  hipStream_t stream = (hipStream_t)stream_;
  const int rp = fold_bwd_rows_per(ntot);
  const unsigned nslab = (unsigned)((ntot + rp - 1) / rp);
  float* part = dg ? (float*)scratch : nullptr;
#define FOLDB(BF)                                                                                                                          \
  hipLaunchKernelGGL(fold_weights_bwd_kernel<BF>, dim3((unsigned)(K / 64), nslab), dim3(256), 0, stream, (const uint16_t*)dwg,                 \
                     (const uint16_t*)w0, (const uint16_t*)w1, (const uint16_t*)w2, (int)n0, (int)n1, (int)n2, ld0, ld1, ld2,              \
                     (const uint16_t*)g, (uint16_t*)dw0, (uint16_t*)dw1, (uint16_t*)dw2, part, (int)K, rp);                                \
  if (dg) hipLaunchKernelGGL(fold_weights_dg_kernel<BF>, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, stream, part, (uint16_t*)dg, (int)K, (int)nslab)
  if (dtype == FAT5_BF16) { FOLDB(true); } else { FOLDB(false); }
#undef FOLDB
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "fold_weights_bwd launch");
  return FAT5_OK;
}

// ---- gated activation (rowwise_kernels.h) ----
static int gated_act_check(const char* what, int64_t rows, int64_t F, int act, int dtype, std::initializer_list<const void*> ptrs,
                           std::initializer_list<int64_t> strides) {
  if (!dtype_ok(dtype)) return fail(FAT5_EINVAL, "%s: bad dtype", what);
  if (act != FAT5_ACT_GELU_TANH && act != FAT5_ACT_RELU) return fail(FAT5_EINVAL, "%s: act %d", what, act);
  const int v = vec_of(dtype);
  if (rows < 0 || F <= 0 || F % v || F > 0x7fffffffLL) return fail(FAT5_EINVAL, "%s: F must be a positive multiple of %d", what, v);
  for (const void* q : ptrs)
    if (!q || !aligned16(q)) return fail(FAT5_EINVAL, "%s: null or unaligned pointer (16-byte aligned bases)", what);
  for (int64_t st : strides)
    if (st % v) return fail(FAT5_EINVAL, "%s: row strides must be multiples of %d elements", what, v);
  return FAT5_OK;
}
int fat5_gated_act_fwd(const void* h0, const void* h1, void* out, int64_t rows, int64_t F, int64_t s0, int64_t s1, int64_t so, int act,
                       int dtype, void* stream_) {
  int rc = gated_act_check("gated_act_fwd", rows, F, act, dtype, {h0, h1, out}, {s0, s1, so});
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int v = vec_of(dtype);
  const unsigned gx = (unsigned)((F / v + 255) / 256);
  const size_t esz = dtype == FAT5_F32 ? 4 : 2;
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {  // (grid.y limit)
    const unsigned gy = (unsigned)std::min<int64_t>(65535, rows - r0);
    const char *a = (const char*)h0 + r0 * s0 * esz, *b = (const char*)h1 + r0 * s1 * esz;
    char* o = (char*)out + r0 * so * esz;
    dispatch_dtype(dtype, [&](auto dt_) {
      constexpr int DT = decltype(dt_)::value;
      if (act == FAT5_ACT_RELU)
        hipLaunchKernelGGL((gated_act_fwd_kernel<DT, FAT5_ACT_RELU>), dim3(gx, gy), dim3(256), 0, stream, a, b, o, (int)F, s0, s1, so);
      else
        hipLaunchKernelGGL((gated_act_fwd_kernel<DT, FAT5_ACT_GELU_TANH>), dim3(gx, gy), dim3(256), 0, stream, a, b, o, (int)F, s0, s1, so);
    });
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "gated_act_fwd launch");
  return FAT5_OK;
}
int fat5_gated_act_bwd(const void* dout, const void* h0, const void* h1, void* dh0, void* dh1, int64_t rows, int64_t F, int64_t sd, int64_t s0,
                       int64_t s1, int64_t sg0, int64_t sg1, int act, int dtype, void* stream_) {
  int rc = gated_act_check("gated_act_bwd", rows, F, act, dtype, {dout, h0, h1, dh0, dh1}, {sd, s0, s1, sg0, sg1});
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int v = vec_of(dtype);
  const unsigned gx = (unsigned)((F / v + 255) / 256);
  const size_t esz = dtype == FAT5_F32 ? 4 : 2;
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
    const unsigned gy = (unsigned)std::min<int64_t>(65535, rows - r0);
    const char *g = (const char*)dout + r0 * sd * esz, *a = (const char*)h0 + r0 * s0 * esz, *b = (const char*)h1 + r0 * s1 * esz;
    char *o0 = (char*)dh0 + r0 * sg0 * esz, *o1 = (char*)dh1 + r0 * sg1 * esz;
    dispatch_dtype(dtype, [&](auto dt_) {
      constexpr int DT = decltype(dt_)::value;
      if (act == FAT5_ACT_RELU)
        hipLaunchKernelGGL((gated_act_bwd_kernel<DT, FAT5_ACT_RELU>), dim3(gx, gy), dim3(256), 0, stream, g, a, b, o0, o1, (int)F, sd, s0, s1, sg0, sg1);
      else
        hipLaunchKernelGGL((gated_act_bwd_kernel<DT, FAT5_ACT_GELU_TANH>), dim3(gx, gy), dim3(256), 0, stream, g, a, b, o0, o1, (int)F, sd, s0, s1, sg0, sg1);
    });
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "gated_act_bwd launch");
  return FAT5_OK;
}

int fat5_rmsnorm_unit_bwd(const void* gy, const void* x, const float* rstd, void* dx, void* xhat, int64_t rows, int64_t n, int64_t gy_stride,
                          int64_t x_stride, int64_t dx_stride, int64_t xhat_stride, const void* dres, int64_t dres_stride, int dtype,
                          void* stream_) {
  if (!gy || !x || !rstd || !dx || !xhat) return fail(FAT5_EINVAL, "rmsnorm_unit_bwd: null pointer");
  if (!dtype_ok(dtype)) return fail(FAT5_EINVAL, "rmsnorm_unit_bwd: bad dtype");
  const int v = vec_of(dtype);
  if (rows <= 0 || n <= 0 || n % v || n > 4 * 64 * v || gy_stride % v || x_stride % v || dx_stride % v || xhat_stride % v || !aligned16(gy) ||
      !aligned16(x) || !aligned16(dx) || !aligned16(xhat) || (dres && (!aligned16(dres) || dres_stride % v)))
    return fail(FAT5_EINVAL, "rmsnorm_unit_bwd: n must be a multiple of %d and at most %d; 16-byte aligned rows", v, 4 * 64 * v);
  hipStream_t stream = (hipStream_t)stream_;
  const int grid = (int)((rows + 3) / 4);
  const int nch = (int)((n + 64 * v - 1) / (64 * v));
  dispatch_dtype(dtype, [&](auto dt_) {
    constexpr int DT = decltype(dt_)::value;
    if (nch <= 2)
      hipLaunchKernelGGL((rmsnorm_unit_bwd_kernel<DT, 2>), dim3(grid), dim3(256), 0, stream, gy, x, rstd, dx, xhat, rows, (int)n, gy_stride, x_stride, dx_stride, xhat_stride, dres, dres_stride);
    else
      hipLaunchKernelGGL((rmsnorm_unit_bwd_kernel<DT, 4>), dim3(grid), dim3(256), 0, stream, gy, x, rstd, dx, xhat, rows, (int)n, gy_stride, x_stride, dx_stride, xhat_stride, dres, dres_stride);
  });
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "rmsnorm_unit_bwd launch");
  return FAT5_OK;
}

// ============================================================================================
// Cross-entropy
// ============================================================================================
#define CE_DISPATCH(...)                         \
  dispatch_dtype(dtype, [&](auto dt_) {            \
    constexpr int DT = decltype(dt_)::value;       \
    __VA_ARGS__                                    \
  });

int fat5_ce_fwd(const void* logits, const int64_t* labels, float* losses, float* z_losses, float* lse, int64_t rows,
                int64_t n_cols, int64_t row_stride, float smoothing, float logit_scale, float lse_square_scale,
                int64_t ignore_index, int use_precomputed_lse, int dtype, void* stream_) {
  if (!logits || !labels || !losses || !z_losses || !lse) return fail(FAT5_EINVAL, "ce_fwd: null pointer");
  if (!dtype_ok(dtype)) return fail(FAT5_EINVAL, "ce_fwd: bad dtype");
  if (rows <= 0 || n_cols <= 0 || n_cols > 0x7fffffffLL || rows > 0x7fffffffLL) return fail(FAT5_EINVAL, "ce_fwd: bad shape");
  hipStream_t stream = (hipStream_t)stream_;
  const int v = vec_of(dtype);
  const bool vecok = (n_cols % v == 0) && (row_stride % v == 0) && aligned16(logits);
  CE_DISPATCH({
    if (vecok)
      hipLaunchKernelGGL((ce_fwd_kernel<DT, true>), dim3((int)rows), dim3(256), 0, stream, logits, labels, losses, z_losses, lse,
                         (int)n_cols, row_stride, smoothing, logit_scale, lse_square_scale, ignore_index, use_precomputed_lse);
    else
      hipLaunchKernelGGL((ce_fwd_kernel<DT, false>), dim3((int)rows), dim3(256), 0, stream, logits, labels, losses, z_losses, lse,
                         (int)n_cols, row_stride, smoothing, logit_scale, lse_square_scale, ignore_index, use_precomputed_lse);
  })
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "ce_fwd launch");
  return FAT5_OK;
}

int fat5_ce_bwd(const float* dlosses, int64_t dloss_stride, const void* logits, const float* lse, const int64_t* labels,
                void* dlogits, int64_t rows, int64_t n_cols, int64_t row_stride, int64_t drow_stride, float smoothing,
                float logit_scale, float lse_square_scale, int64_t ignore_index, int dtype, void* stream_) {
  if (!dlosses || !logits || !lse || !labels || !dlogits) return fail(FAT5_EINVAL, "ce_bwd: null pointer");
  if (!dtype_ok(dtype)) return fail(FAT5_EINVAL, "ce_bwd: bad dtype");
  if (rows <= 0 || n_cols <= 0 || n_cols > 0x7fffffffLL || rows > 0x7fffffffLL) return fail(FAT5_EINVAL, "ce_bwd: bad shape");
  hipStream_t stream = (hipStream_t)stream_;
  const int v = vec_of(dtype);
  const bool vecok = (n_cols % v == 0) && (row_stride % v == 0) && (drow_stride % v == 0) && aligned16(logits) && aligned16(dlogits);
  const int per_block = 256 * v;
  dim3 grid((unsigned)rows, (unsigned)((n_cols + per_block - 1) / per_block));
  if (grid.y > 65535) return fail(FAT5_EINVAL, "ce_bwd: too many columns");
  CE_DISPATCH({
    if (vecok)
      hipLaunchKernelGGL((ce_bwd_kernel<DT, true>), grid, dim3(256), 0, stream, dlosses, dloss_stride, logits, lse, labels, dlogits,
                         (int)n_cols, row_stride, drow_stride, smoothing, logit_scale, lse_square_scale, ignore_index);
    else
      hipLaunchKernelGGL((ce_bwd_kernel<DT, false>), grid, dim3(256), 0, stream, dlosses, dloss_stride, logits, lse, labels, dlogits,
                         (int)n_cols, row_stride, drow_stride, smoothing, logit_scale, lse_square_scale, ignore_index);
  })
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "ce_bwd launch");
  return FAT5_OK;
}

int fat5_ce_fwd_bwd(const void* logits, const int64_t* labels, const float* dlosses, int64_t dloss_stride, float* losses, float* z_losses,
                    float* lse, void* dlogits, int64_t rows, int64_t n_cols, int64_t row_stride, int64_t drow_stride, float smoothing,
                    float logit_scale, float lse_square_scale, int64_t ignore_index, int dtype, void* stream_) {
  if (!logits || !labels || !dlosses || !losses || !z_losses || !lse || !dlogits) return fail(FAT5_EINVAL, "ce_fwd_bwd: null pointer");
  if (!dtype_ok(dtype)) return fail(FAT5_EINVAL, "ce_fwd_bwd: bad dtype");
  if (rows <= 0 || n_cols <= 0 || n_cols > 0x7fffffffLL || rows > 0x7fffffffLL) return fail(FAT5_EINVAL, "ce_fwd_bwd: bad shape");
  const int v = vec_of(dtype);
  const bool vecok = (n_cols % v == 0) && (row_stride % v == 0) && (drow_stride % v == 0) && aligned16(logits) && aligned16(dlogits);
  if (!vecok) {  // the two launches: same results
    const int rc = fat5_ce_fwd(logits, labels, losses, z_losses, lse, rows, n_cols, row_stride, smoothing, logit_scale, lse_square_scale,
                               ignore_index, 0, dtype, stream_);
    if (rc != FAT5_OK) return rc;
    return fat5_ce_bwd(dlosses, dloss_stride, logits, lse, labels, dlogits, rows, n_cols, row_stride, drow_stride, smoothing, logit_scale,
                       lse_square_scale, ignore_index, dtype, stream_);
  }
  hipStream_t stream = (hipStream_t)stream_;
  const bool hold = n_cols <= (int64_t)16 * 256 * v;
  CE_DISPATCH({
    if (hold)
      hipLaunchKernelGGL((ce_fwd_bwd_kernel<DT, true>), dim3((int)rows), dim3(256), 0, stream, logits, labels, dlosses, dloss_stride, losses,
                         z_losses, lse, dlogits, (int)n_cols, row_stride, drow_stride, smoothing, logit_scale, lse_square_scale, ignore_index);
    else
      hipLaunchKernelGGL((ce_fwd_bwd_kernel<DT, false>), dim3((int)rows), dim3(256), 0, stream, logits, labels, dlosses, dloss_stride, losses,
                         z_losses, lse, dlogits, (int)n_cols, row_stride, drow_stride, smoothing, logit_scale, lse_square_scale, ignore_index);
  })
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "ce_fwd_bwd launch");
  return FAT5_OK;
}


// ============================================================================================
// AdamWScale
// ============================================================================================
size_t fat5_sizeof_adamw_tensor(void) { return sizeof(fat5_adamw_tensor); }

static int adamw_step_impl(const fat5_adamw_tensor* table, int32_t n_tensors, int32_t n_chunks, float* partials, double lr_, double beta1_,
                           double beta2_, double weight_decay_, double eps_, int dtype, int state_dtype, int flags, const float* grad_coef,
                           const float* dev_scalars, void* stream_) {
  // scalars reach the kernels as the fp32 "opmath" values the reference's ops see: each Python double is cast once
  const float beta1 = (float)beta1_, beta2 = (float)beta2_, eps = (float)eps_;
  const float a1 = (float)(1.0 - beta1_), a2 = (float)(1.0 - beta2_);
  const float wdf = weight_decay_ > 0.0 ? (float)(-lr_ * weight_decay_) : 0.f;  // (the reference decays only `if weight_decay > 0.0`, :209)
  const float lr_small = (float)(lr_ * 1e-3);
  const int kahan = flags & FAT5_ADAMW_KAHAN, plain = (flags & FAT5_ADAMW_PLAIN_STEP) ? 1 : 0;
  if (!table || !partials) return fail(FAT5_EINVAL, "adamw: null table / partials");
  if (n_tensors <= 0 || n_chunks <= 0) return fail(FAT5_EINVAL, "adamw: empty group");
  if (!dtype_ok(dtype) || !dtype_ok(state_dtype)) return fail(FAT5_EINVAL, "adamw: bad dtype");
  if (state_dtype == FAT5_F32 && dtype != FAT5_F32)
    return fail(FAT5_EINVAL, "adamw: fp32 states beside 16-bit parameters do not exist in the reference (use_state_dtype is fp16 / bf16, :101-103)");
  if (kahan && dtype == FAT5_F32) return fail(FAT5_EINVAL, "adamw: Kahan compensation is for 16-bit parameters (reference :107-113)");
  hipStream_t stream = (hipStream_t)stream_;
  dispatch_dtype(dtype, [&](auto dt_) {
    constexpr int DT = decltype(dt_)::value;
    hipLaunchKernelGGL((adamw_sumsq_kernel<DT, false>), dim3(n_chunks), dim3(256), 0, stream, table, n_tensors, partials);
    dispatch_dtype(state_dtype, [&](auto st_) {
      constexpr int SDT = decltype(st_)::value;
      if constexpr (SDT != FAT5_F32 || DT == FAT5_F32) {
        if constexpr (DT != FAT5_F32) {
          if (kahan) {
            hipLaunchKernelGGL((adamw_update_kernel<DT, SDT, true>), dim3(n_chunks), dim3(256), 0, stream, table, n_tensors, partials, beta1,
                               beta2, a1, a2, wdf, eps, grad_coef, plain, lr_small, dev_scalars);
            return;
          }
        }
        hipLaunchKernelGGL((adamw_update_kernel<DT, SDT, false>), dim3(n_chunks), dim3(256), 0, stream, table, n_tensors, partials, beta1,
                           beta2, a1, a2, wdf, eps, grad_coef, plain, lr_small, dev_scalars);
      }
    });
  });
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "adamw launch");
  return FAT5_OK;
}
int fat5_adamw_scale_step(const fat5_adamw_tensor* table, int32_t n_tensors, int32_t n_chunks, float* partials, double lr, double beta1,
                          double beta2, double weight_decay, double eps, int dtype, int state_dtype, int flags, void* stream) {
  return adamw_step_impl(table, n_tensors, n_chunks, partials, lr, beta1, beta2, weight_decay, eps, dtype, state_dtype, flags, nullptr, nullptr, stream);
}
int fat5_adamw_scale_step_clipped(const fat5_adamw_tensor* table, int32_t n_tensors, int32_t n_chunks, float* partials, double lr,
                                  double beta1, double beta2, double weight_decay, double eps, int dtype, int state_dtype, int flags,
                                  const float* grad_coef, void* stream) {
  if (!grad_coef) return fail(FAT5_EINVAL, "adamw: grad_coef is NULL");
  return adamw_step_impl(table, n_tensors, n_chunks, partials, lr, beta1, beta2, weight_decay, eps, dtype, state_dtype, flags, grad_coef, nullptr, stream);
}
int fat5_adamw_scale_step_dev(const fat5_adamw_tensor* table, int32_t n_tensors, int32_t n_chunks, float* partials, const float* dev_scalars,
                              double beta1, double beta2, double eps, int dtype, int state_dtype, int flags, const float* grad_coef,
                              void* stream) {
  if (!dev_scalars) return fail(FAT5_EINVAL, "adamw: dev_scalars is NULL");
  return adamw_step_impl(table, n_tensors, n_chunks, partials, 0.0, beta1, beta2, 0.0, eps, dtype, state_dtype, flags, grad_coef, dev_scalars, stream);
}
int fat5_adamw_grad_sumsq(const fat5_adamw_tensor* table, int32_t n_tensors, int32_t n_chunks, float* partials, int dtype, void* stream_) {
  if (!table || !partials) return fail(FAT5_EINVAL, "adamw: null table / partials");
  if (n_tensors <= 0 || n_chunks <= 0) return fail(FAT5_EINVAL, "adamw: empty group");
  if (!dtype_ok(dtype)) return fail(FAT5_EINVAL, "adamw: bad dtype");
  hipStream_t stream = (hipStream_t)stream_;
  dispatch_dtype(dtype, [&](auto dt_) {
    constexpr int DT = decltype(dt_)::value;
    hipLaunchKernelGGL((adamw_sumsq_kernel<DT, true>), dim3(n_chunks), dim3(256), 0, stream, table, n_tensors, partials);
  });
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "adamw grad sumsq launch");
  return FAT5_OK;
}

}  // extern "C"
