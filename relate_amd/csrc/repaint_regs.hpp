// Re-painting of one window from its stepping stones (FastPainting::RePaintSection,
// src/fast_painting.cpp:654-1126) for all prepared targets, with the register-resident sweeps of
// paint_regs.hpp: the section of target t is the piece [bbj, bej] of its derived-site list, seeded with the
// alpha stone at its first site and the beta stone at its last, and the product alpha * beta of every site
// of the section is written as a row of the `topology` matrix (float), with its logscale.
//
// Two launches per window.  The reference keeps every alpha row of a section (double, :786-919) until the
// backward pass has multiplied it into beta (:958-969,:1067-1078, before the beta row is rescaled, :1081-1095).
// Here the forward launch keeps every B-th alpha row (a checkpoint, with the sum that goes with it) in a scratch
// matrix in HBM; the backward launch walks the section in blocks of B rows from its end: it parks beta in
// accumulation registers (AGPRs), loads the block's checkpoint, runs the forward recursion again over the block
// -- the same instructions as the forward launch, hence the same doubles -- parking each alpha row in AGPRs too,
// takes beta back and walks the block downwards, multiplying.  Per element of the topology that is 16/B bytes of
// checkpoint traffic beside the 4 bytes of the result, instead of 16.  B follows from the registers: the AGPR
// slots (2*MPL registers each, B - 1 alpha rows and beta) are what 256 registers per lane (two waves per SIMD) leave
// beside the kernel's own and the compiler's AGPRs, at most 8 rows; from 32 words per lane on there is room for one
// slot only and every row is a checkpoint (B = 1: the reference's scheme).  Rows have a pitch of G*MPL*64 elements (>= N), so the lanes past N and the
// target's own lane, which hold exact zeros, are stored like any other.
//
// Internal header; instantiated per G by paint_regs_g*.hip.
#pragma once
#include "paint_regs.hpp"

namespace relate {
namespace regs {

constexpr int repaint_block(int MPL) { return repaint_block_rows(MPL); }  // paint_kernels.h: the host sizes the scratch with it

// A kernel whose asm names AGPRs shares its register budget with them: the backward kernel is compiled with
// amdgpu_num_vgpr(32), which gives the compiler v0..v31 and a0..a31 (it parks spilled values there); the slots
// start at a32, and slots, compiler AGPRs and the 56 + 2*MPL registers of the kernel stay inside 256 per lane: two
// waves per SIMD (at one, the launch is bound by the latency of its dependent chains).  Every AGPR the asm names must be on a clobber list (the back end sizes the
// register file from them), and tools/check_regs.py reads the "relate_limits a32" left in the assembly and fails
// the build if a compiler instruction names an AGPR from a32 on (or, as everywhere, a VGPR from v48 on).
constexpr int kSlot0 = 32;
#define RELATE_BWD_REPAINT_ATTR __attribute__((amdgpu_num_vgpr(32), amdgpu_num_sgpr(RELATE_CS_ATTR)))
#define RELATE_ACLOB_64 "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"
#define RELATE_ACLOB_96 RELATE_ACLOB_64, "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define RELATE_ACLOB_112 RELATE_ACLOB_96, "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143"
#define RELATE_ACLOB_120 RELATE_ACLOB_112, "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151"
#define RELATE_ACLOB_128 RELATE_ACLOB_120, "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159"
#define RELATE_ACLOB_144 RELATE_ACLOB_128, "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175"
template <int MPL>
__device__ __forceinline__ void claim_slots()
{
  constexpr int n = repaint_block(MPL) > 1 ? 2 * MPL * repaint_block(MPL) : 0;
  static_assert(kSlot0 + n <= 256 && kA0 + 2 * MPL + kSlot0 + n <= 256, "registers of two waves per SIMD");
  // no slots (every row a checkpoint): one AGPR is named all the same, so that the register budget of the kernel is
  // split the same way for every instantiation (without any AGPR the back end hands all of it to the compiler as VGPRs)
  if constexpr (n == 0) asm volatile("; relate_limits a%0" ::"n"(kSlot0) : "a32");
  if constexpr (n == 64) asm volatile("; relate_limits a%0" ::"n"(kSlot0) : RELATE_ACLOB_64);
  if constexpr (n == 96) asm volatile("; relate_limits a%0" ::"n"(kSlot0) : RELATE_ACLOB_96);
  if constexpr (n == 112) asm volatile("; relate_limits a%0" ::"n"(kSlot0) : RELATE_ACLOB_112);
  if constexpr (n == 120) asm volatile("; relate_limits a%0" ::"n"(kSlot0) : RELATE_ACLOB_120);
  if constexpr (n == 128) asm volatile("; relate_limits a%0" ::"n"(kSlot0) : RELATE_ACLOB_128);
  if constexpr (n == 144) asm volatile("; relate_limits a%0" ::"n"(kSlot0) : RELATE_ACLOB_144);
  static_assert(n == 0 || n == 64 || n == 96 || n == 112 || n == 120 || n == 128 || n == 144, "an AGPR clobber list for every block size");
}
// state block <-> AGPR slot S (registers a[kSlot0 + S*2*MPL .. kSlot0 + (S+1)*2*MPL))
template <int S, int MPL>
__device__ __forceinline__ void slot_save()
{
  static_for<0, 2 * MPL>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    asm volatile("v_accvgpr_write_b32 a[%0], v[%1]" ::"n"(kSlot0 + S * 2 * MPL + i), "n"(kA0 + i));
  });
}
template <int S, int MPL>
__device__ __forceinline__ void slot_load()
{
  asm volatile("s_nop 1");
  static_for<0, 2 * MPL>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    asm volatile("v_accvgpr_read_b32 v[%1], a[%0]" ::"n"(kSlot0 + S * 2 * MPL + i), "n"(kA0 + i));
  });
}
// float(word I of slot S * state word I)
template <int S, int I, int MPL>
__device__ __forceinline__ float slot_times_state()
{
  float f;
  asm volatile(
      "v_accvgpr_read_b32 v[%[t]], a[%[a]]\n\t"
      "v_accvgpr_read_b32 v[%[t]+1], a[%[a]+1]\n\t"
      "v_mul_f64 v[%[t]:%[t]+1], v[%[t]:%[t]+1], v[%[s]:%[s]+1]\n\t"
      "v_cvt_f32_f64 %[f], v[%[t]:%[t]+1]"
      : [f] "=v"(f)
      : [a] "n"(kSlot0 + S * 2 * MPL + 2 * I), [t] "n"(kT), [s] "n"(kA0 + 2 * I));
  return f;
}

struct RepaintKArgs {
  const uint64_t* rows;
  const int32_t* dk_all;
  const SiteRec* rec_all;
  const int64_t* off;
  const int32_t* nd_all;
  const int32_t* bbj_all;   // [K][W] list index of the alpha stone of window w
  const int32_t* bej_all;   // [K][W] list index of the beta stone
  const int32_t* order;     // [K] launch order
  const uint8_t* stones;    // [K] records of this window (56+8N bytes each), target t at t*(56+8N)
  const int64_t* row_off;   // [K] first row of target t in scratch / topology / logscales
  const double* last_coef;  // [K] gap terms of the section's last site: r[last] alone (:737-750)
  const double* last_nx;
  double* scratch;          // [rows / B + K + 1][pitch] alpha checkpoints: row j of target t at row_off[t] / B + t + j / B
  double* cp_sum;           // the sum of each checkpoint row as the forward recursion holds it (1 after a rescale)
  float* topology;          // [rows][pitch]
  float* logscales;         // [rows]
  int w, pitch;
  Scal p;
};
// Pointers that went through an empty asm (to keep the compiler from fetching them again inside a loop) have
// lost their provenance; without it the accesses become FLAT instructions, whose waits include the scalar-load
// counter and with it the row fetch in flight.  These casts say "global memory" again.
typedef __attribute__((address_space(1))) double GlobalDouble;
typedef __attribute__((address_space(1))) float GlobalFloat;
__device__ __forceinline__ GlobalDouble* as_global(double* p) { return (GlobalDouble*)p; }
__device__ __forceinline__ const GlobalDouble* as_global(const double* p) { return (const GlobalDouble*)p; }
__device__ __forceinline__ GlobalFloat* as_global(float* p) { return (GlobalFloat*)p; }

// Element of a row for this lane: `row` is the same in every lane (it lives in scalar registers), `lane_bytes` the
// lane's offset in it; the access becomes scalar base + 32-bit vector offset, and no lane keeps a 64-bit pointer.
template <class T>
__device__ __forceinline__ T* at_lane(T* row, uint32_t lane_bytes)
{
  // (opaque at every use: hoisted out of a loop, the zero-extended offset would become a 64-bit register pair of its
  // own and the access a vector-address one)
  asm volatile("" : "+v"(lane_bytes));
  return (T*)((__attribute__((address_space(1))) char*)row + lane_bytes);
}
// One float through the scalar cache into s[kSkEdge] (free in these kernels: their edge masks travel as arguments).
// Ask for it before a sweep and take it after: the sweep's first wait for its row words covers it.  A vector load
// here would be waited for together with every store of the row before -- a round trip to memory per row.
__device__ __forceinline__ void sload_float(const float* p) { asm volatile("s_load_dword s[%1], %0, 0x0" ::"s"(p), "n"(kSkEdge) : "memory"); }
__device__ __forceinline__ float sloaded_float()
{
  uint32_t x;
  asm volatile("s_mov_b32 %0, s[%1]" : "=s"(x) : "n"(kSkEdge));
  return __uint_as_float(x);
}

// Row of doubles -> state block without a compiler register in between: word i of this lane from the uniform pointer
// `row` (scalar base, the lane's byte offset, word offsets as immediates; a new base every 8 words keeps them inside
// the 12 bits of the instruction).  The caller waits (state_loads_wait) once every load of the row is on its way.
template <int MPL>
__device__ __forceinline__ void state_load_row(const GlobalDouble* row, uint32_t lane_bytes8)
{
  static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const GlobalDouble* base = row + (size_t)(i / 8) * 512;
    const uint32_t o = lane_bytes8;  // (a lambda's asm operands must be its own variables)
    asm volatile("global_load_dwordx2 v[%0:%0+1], %1, %2 offset:%3" ::"n"(kA0 + 2 * i), "v"(o), "s"(base), "n"((i % 8) * 512) : "memory");
  });
}
__device__ __forceinline__ void state_loads_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The state block parked in LDS and multiplied from there: word i of lane l of wave w at park + ((w*MPL + i)*64 + l)*8
// (every lane reads back what it wrote itself: no barrier).  parked_product leaves float(parked word * state word) of
// every word in the row `row`; the reads go two at a time through v[kX], v[kT] with a full wait each (a scalar load
// may be in flight, and those return in no order, so a counted wait would prove nothing).
template <int MPL>
__device__ __forceinline__ void park_state(uint32_t park_addr)
{
  static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const uint32_t a = park_addr;
    asm volatile("ds_write_b64 %0, v[%1:%1+1] offset:%2" ::"v"(a), "n"(kA0 + 2 * i), "n"(i * 512) : "memory");
  });
}
template <int MPL>
__device__ __forceinline__ void parked_product(uint32_t park_addr, GlobalFloat* row, uint32_t lane_bytes)
{
  static_for<0, (MPL + 1) / 2>([&](auto pc) __attribute__((always_inline)) {
    constexpr int i = 2 * decltype(pc)::value;
    float f0, f1;
    const uint32_t pa = park_addr, lb = lane_bytes;
    GlobalFloat* const rw = row;
    if constexpr (i + 1 < MPL) {
      asm volatile(
          "ds_read_b64 v[%[x]:%[x]+1], %[a] offset:%[o0]\n\t"
          "ds_read_b64 v[%[t]:%[t]+1], %[a] offset:%[o1]\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_mul_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], v[%[s0]:%[s0]+1]\n\t"
          "v_mul_f64 v[%[t]:%[t]+1], v[%[t]:%[t]+1], v[%[s1]:%[s1]+1]\n\t"
          "v_cvt_f32_f64 %[f0], v[%[x]:%[x]+1]\n\t"
          "v_cvt_f32_f64 %[f1], v[%[t]:%[t]+1]"
          : [f0] "=&v"(f0), [f1] "=&v"(f1)
          : [a] "v"(pa), [x] "n"(kX), [t] "n"(kT), [o0] "n"(i * 512), [o1] "n"((i + 1) * 512), [s0] "n"(kA0 + 2 * i),
            [s1] "n"(kA0 + 2 * i + 2)
          : "memory");
      at_lane(rw, lb)[(size_t)i * 64] = f0;
      at_lane(rw, lb)[(size_t)(i + 1) * 64] = f1;
    } else {
      asm volatile(
          "ds_read_b64 v[%[x]:%[x]+1], %[a] offset:%[o0]\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_mul_f64 v[%[x]:%[x]+1], v[%[x]:%[x]+1], v[%[s0]:%[s0]+1]\n\t"
          "v_cvt_f32_f64 %[f0], v[%[x]:%[x]+1]"
          : [f0] "=&v"(f0)
          : [a] "v"(pa), [x] "n"(kX), [o0] "n"(i * 512), [s0] "n"(kA0 + 2 * i)
          : "memory");
      at_lane(rw, lb)[(size_t)i * 64] = f0;
    }
  });
}
// What a block of the backward launch needs from the lists, through the scalar cache and with one wait for all of
// it (as vector loads, each waits for every store issued before it, and there are a dozen of them).  The memory
// behind these pointers was written by earlier launches.
typedef uint32_t __attribute__((ext_vector_type(4))) Quad;
struct BlockSeeds {
  int dk_prev, dk_cur, dk_next;  // list entries around the first backward step
  int dk_f1, dk_f2;              // the first two sites of the forward run
  double coef_top, nx_top;       // record after the first backward step's site
  double coef_f, nx_f;           // record of the block's first row
  double sum;                    // the checkpoint's sum
};
__device__ __forceinline__ BlockSeeds load_block_seeds(const int32_t* p_prev, const int32_t* p_cur, const int32_t* p_next, const int32_t* p_f1,
                                                       const int32_t* p_f2, const SiteRec* r_top, const SiteRec* r_f, const double* p_sum)
{
  uint32_t a, b, c, d, e;
  Quad qt, qf;
  uint64_t sm;
  asm volatile(
      "s_load_dword %0, %8, 0x0\n\t"
      "s_load_dword %1, %9, 0x0\n\t"
      "s_load_dword %2, %10, 0x0\n\t"
      "s_load_dword %3, %11, 0x0\n\t"
      "s_load_dword %4, %12, 0x0\n\t"
      "s_load_dwordx4 %5, %13, 0x0\n\t"
      "s_load_dwordx4 %6, %14, 0x0\n\t"
      "s_load_dwordx2 %7, %15, 0x0\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d), "=&s"(e), "=&s"(qt), "=&s"(qf), "=&s"(sm)
      : "s"(p_prev), "s"(p_cur), "s"(p_next), "s"(p_f1), "s"(p_f2), "s"(r_top), "s"(r_f), "s"(p_sum)
      : "memory");
  BlockSeeds r;
  r.dk_prev = (int)a;
  r.dk_cur = (int)b;
  r.dk_next = (int)c;
  r.dk_f1 = (int)d;
  r.dk_f2 = (int)e;
  r.coef_top = __hiloint2double((int)qt.y, (int)qt.x);
  r.nx_top = __hiloint2double((int)qt.w, (int)qt.z);
  r.coef_f = __hiloint2double((int)qf.y, (int)qf.x);
  r.nx_f = __hiloint2double((int)qf.w, (int)qf.z);
  r.sum = __longlong_as_double((long long)sm);
  return r;
}

// S < lo || S > hi with the thresholds fetched where they are needed (rare: see surely_in_range); the loops of
// several waves per target use this, because their exchange passes through the registers that hold the thresholds
// for out_of_range().
typedef const __attribute__((address_space(4))) RepaintKArgs* RepaintKArgsPtr;
__device__ __forceinline__ bool repaint_out_of_range_exact(double S)
{
  RepaintKArgsPtr q = (RepaintKArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(q));
  return __builtin_amdgcn_ballot_w64(S < q->p.c.lo || S > q->p.c.hi) != 0;  // S is the same in every lane
}

template <int I>
__device__ __forceinline__ double state_get()
{
  double v;
  asm volatile("v_mov_b64 %0, v[%1:%1+1]" : "=v"(v) : "n"(kA0 + 2 * I));
  return v;
}

// The same with the slot chosen at run time (a wave-uniform index below NS): the sweeps around these calls then
// exist once in the code instead of once per row of a block, which keeps the kernel inside the instruction cache.
template <int MPL, int NS>
__device__ __forceinline__ void slot_save_at(int s)
{
  static_for<0, NS>([&](auto sc) __attribute__((always_inline)) {
    if (s == decltype(sc)::value) slot_save<decltype(sc)::value, MPL>();
  });
}
// row `row` (uniform; this lane's column at `lane_bytes`, words 64 floats apart) = float(slot s * state block)
template <int MPL, int NS>
__device__ __forceinline__ void slot_product_at(int s, GlobalFloat* row, uint32_t lane_bytes)
{
  static_for<0, NS>([&](auto sc) __attribute__((always_inline)) {
    if (s == decltype(sc)::value) {
      static_for<0, MPL>([&](auto wc) __attribute__((always_inline)) {
        constexpr int wd = decltype(wc)::value;
        at_lane(row, lane_bytes)[(size_t)wd * 64] = slot_times_state<decltype(sc)::value, wd, MPL>();
      });
    }
  });
}

template <int MPL, int G, bool BWD>
__device__ __forceinline__ void repaint_body(const RepaintKArgs& ka)
{
  static_assert(MPL >= 4 && MPL <= kMaxMPL, "states per lane");
  extern __shared__ double lds_dyn[];
  double (*red)[G] = reinterpret_cast<double (*)[G]>(lds_dyn);
  claim_registers<MPL>();
  int slot = 0;
  const int rank = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 3;
  const int t = __builtin_amdgcn_readfirstlane(ka.order[blockIdx.x]);
  const int k = ka.p.k_begin + t;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wbase = wave * MPL;
  const int stride = ka.p.row_stride;
  const int N = ka.p.N;
  const long long o = ka.off[t];
  const int32_t* __restrict__ dk = ka.dk_all + o;
  const SiteRec* __restrict__ rec = ka.rec_all + o;
  const int nd_chunk = __builtin_amdgcn_readfirstlane(ka.nd_all[t]);
  const int jb = __builtin_amdgcn_readfirstlane(ka.bbj_all[(long long)t * ka.p.W + ka.w]);
  const int je = __builtin_amdgcn_readfirstlane(ka.bej_all[(long long)t * ka.p.W + ka.w]);
  const int nds = je - jb + 1;  // sites of the section (>= 2: the two stones sit on different sites)
  const int kw = k >> 6, kb = k & 63;
  const uint64_t* rows_w = ka.rows + wbase;
  const size_t R = 56 + (size_t)8 * N;
  const uint8_t* stone = ka.stones + (size_t)t * R;
  const size_t row0 = (size_t)sgpr64((uint64_t)ka.row_off[t]);  // uniform, and wanted in scalar registers below
  const size_t pitch = (size_t)ka.pitch;
  const size_t col = (size_t)wbase * 64 + lane;

  auto stateless = [&]() __attribute__((always_inline)) {
    const int full_words = N >> 6, rem = N & 63;
    const bool self_here = kw >= wbase && kw < wbase + MPL;
    const int w_ph = full_words > wbase ? full_words : wbase;
    set_stateless(sgpr64(self_here ? (1ull << kb) : 0ull), sgpr32(self_here ? 2u * (uint32_t)(kw - wbase) : 0u),
                  sgpr64((w_ph == full_words && rem) ? ~((1ull << rem) - 1ull) : ~0ull),
                  sgpr32(w_ph - wbase < MPL ? 2u * (uint32_t)(w_ph - wbase) : 2u * MPL), 2u * MPL);
  };

  const int s_first = __builtin_amdgcn_readfirstlane(dk[jb]);
  const int s_last = __builtin_amdgcn_readfirstlane(dk[je]);
  // the target's own allele at the two ends of the section; it is '1' at every site in between
  const uint64_t sk_first = sgpr64(((ka.rows[(size_t)s_first * stride + kw] >> kb) & 1ull) ? ~0ull : 0ull);
  const uint64_t sk_last = sgpr64(((ka.rows[(size_t)s_last * stride + kw] >> kb) & 1ull) ? ~0ull : 0ull);
  const double rho = ka.p.c.rho_theta;

  constexpr int B = repaint_block(MPL);
  const size_t cp0 = row0 / (size_t)B + (size_t)t;  // first checkpoint row of this target
  // One site of the forward recursion (:829-919) on the state block, shared by the forward launch and by the
  // backward launch's second run over a block: the same instructions in the same order, so that both produce the
  // same doubles.  Returns true when the row was rescaled (lg = log of the sum it was divided by).
  // Scalars that travel: coef_j (gap before this site), the running record pointer, s_next (site after this one).
  //
  // Several waves per target (G > 1) run the site as the recursion kernel's loops of several waves do
  // (paint_regs.hpp): the gap terms and the list entry of the site after the next are used in the scalar registers
  // where their fetch leaves them (set_site_terms before the first site; the fetch itself travels with the next
  // site's row words), the block sum goes through exchange_sum (`step` = j; a barrier before the first site of a run
  // and after its last, where the entries change hands with block_sum_hi), the range check is the scalar one, and a
  // wave without lanes to reset skips that step.  Then coef_j, nx_j and s_next are not used; rp_nx is the row of
  // the site after this one.
  uint32_t wr_addr = 0, rd_base = 0;
  uint32_t stride_bytes = 8u * (uint32_t)stride;
  if constexpr (G > 1) {
    exchange_addresses<G>(red, wr_addr, rd_base);
    set_rows_base(rows_w);
    asm volatile("" : "+s"(stride_bytes));
  }
  auto forward_site = [&](int j, double& S, double& coef_j, double& nx_j, const uint64_t*& rp, const uint64_t*& rp_nx, int& s_next,
                          double rho_l, int stride_l, double& lg) __attribute__((always_inline)) -> bool {
    bool out_of_band;
    if constexpr (G == 1) {
      const double rx = S * coef_j;
      write_stateless<MPL, false>(S, coef_j);
      prefetch_site<true>();
      const uint64_t skm = edge_select(j, nds - 1, sk_last);
      double c0, c1;
      const uint64_t* rp_next = rows_w + (size_t)s_next * stride_l;
      sweep_row<MPL>(c0, c1, rx, rho_l, skm, rp, rp_next);
      S = block_sum_regs<G>(red, slot, c0, c1);
      int s_next2;
      fetched_site<true>(s_next2, coef_j, nx_j);
      out_of_band = out_of_range(S);
      s_next = s_next2;
      rp = rp_next;
    } else {
      const double rx = times_site_coef(S);
      write_stateless_site<MPL>(S);
      const uint64_t skm = edge_select(j, nds - 1, sk_last);
      double c0, c1;
      sweep_row<MPL, 1>(c0, c1, rx, rho_l, skm, rp, rp_nx);
      S = exchange_sum<G>(wr_addr, rd_base, j, c0, c1);
      const uint32_t S_high = sgpr32((uint32_t)__double2hiint(S));
      const uint64_t* rp_nx2 = fetched_row_pointer<true>(stride_bytes);
      out_of_band = !surely_in_range(S_high) && repaint_out_of_range_exact(S);
      rp = rp_nx;
      rp_nx = rp_nx2;
    }
    bool rescaled = false;
    if (out_of_band) {
      const double inv = 1.0 / S;
      static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
        const double f = inv;
        state_scale<decltype(ic)::value>(f);
      });
      lg = log(S);
      S = 1.0;
      rescaled = true;
    }
    return rescaled;
  };

  if constexpr (!BWD) {
    // ---------------- forward (:786-919) ----------------
    const float* __restrict__ alpha0 = reinterpret_cast<const float*>(stone + 32);
    double prev = (double)*reinterpret_cast<const float*>(stone + 8 + 20);  // logscale of the alpha stone
    double acc = 0.0;
    double* arow = ka.scratch + cp0 * pitch + col;
    static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      const int n = (wbase + i) * 64 + lane;
      const double a = (n < N && n != k) ? (double)alpha0[n] : 0.0;
      state_set<i>(a);
      arow[(size_t)i * 64] = a;
      acc += a;
    });
    double S = block_sum_regs<G>(red, slot, acc);
    if (threadIdx.x == 0) {
      ka.logscales[row0] = (float)prev;
      ka.cp_sum[cp0] = S;
    }
    stateless();
    set_thresholds(ka.p.c.lo, ka.p.c.hi);
    int s_cur = __builtin_amdgcn_readfirstlane(dk[jb + 1]);
    int s_next = __builtin_amdgcn_readfirstlane(dk[jb + 2 < nd_chunk ? jb + 2 : nd_chunk - 1]);
    double coef_j = rec[jb].coef, nx_j = rec[jb].nx;
    set_site_pointer(rec + jb + 1);
    // what every site needs from the argument segment, fetched once and opaque to the compiler: a scalar load
    // inside the loop would bring a wait for the row fetch in flight with it
    double* scratch_rows = ka.scratch + cp0 * pitch + col;
    double* sums = ka.cp_sum + cp0;
    float* ls_rows = ka.logscales + row0;
    size_t pitch_l = pitch;
    int stride_l = stride;
    asm volatile("" : "+s"(pitch_l), "+s"(stride_l), "+s"(ls_rows), "+s"(sums));
    asm volatile("" : "+v"(scratch_rows));
    double rho_l = rho;
    asm volatile("" : "+v"(rho_l));
    const uint64_t* rp = rows_w + (size_t)s_cur * stride_l;
    const uint64_t* rp_nx = rows_w + (size_t)s_next * stride_l;
    sload_round<0, RowRounds<MPL>::size(0)>(rp);
    if constexpr (G > 1) {
      set_site_terms(sgpr_f64(coef_j), sgpr_f64(nx_j));
      __syncthreads();  // every wave has read the entries of the sum above
    }
    for (int j = 1; j < nds; j++) {
      if ((j & 31) == 0) rotate_priority(rank, j, 0);
      // the gap before site j (the terms are replaced by the next gap's inside forward_site)
      if constexpr (G == 1) prev += nx_j;
      else add_site_nx(prev);
      float lsj = (float)prev;  // (:839-841) the float is what the rescale term is added to
      double lg;
      if (forward_site(j, S, coef_j, nx_j, rp, rp_nx, s_next, rho_l, stride_l, lg)) {
        prev += lg;
        lsj = (float)((double)lsj + lg);
      }
      if (threadIdx.x == 0) as_global(ls_rows)[j] = lsj;
      if (j % B == 0) {
        GlobalDouble* __restrict__ ar = as_global(scratch_rows + (size_t)(j / B) * pitch_l);
        static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          ar[(size_t)i * 64] = state_get<i>();
        });
        if (threadIdx.x == 0) as_global(sums)[j / B] = S;
      }
    }
    swait();
  } else {
    // ---------------- backward (:921-1107) ----------------
    claim_slots<MPL>();
    const double theta_d = ka.p.c.theta_d, ntheta_d = ka.p.c.ntheta_d;
    const float* __restrict__ beta0 = reinterpret_cast<const float*>(stone + 56 + (size_t)4 * N);
    const float ls_beta = *reinterpret_cast<const float*>(stone + 32 + (size_t)4 * N + 20);
    double prev = (double)ls_beta;
    const uint64_t* row_last = rows_w + (size_t)s_last * stride;
    stateless();
    set_thresholds(ka.p.c.lo, ka.p.c.hi);
    const double inv_nt = ka.p.c.inv_ntheta_d, inv_t = ka.p.c.inv_theta_d;
    const double w_all = ntheta_d, w_mis = theta_d - ntheta_d;
    // rows of the checkpoints and of the topology as uniform pointers plus this lane's byte offset in a float row
    const double* scratch_rows = ka.scratch + cp0 * pitch;
    const double* sums = ka.cp_sum + cp0;
    float* top_rows = ka.topology + row0 * pitch;
    float* ls_rows = ka.logscales + row0;
    size_t pitch_l = pitch;
    int stride_l = stride;
    uint32_t col4 = 4u * (uint32_t)col;
    asm volatile("" : "+s"(pitch_l), "+s"(stride_l), "+s"(ls_rows), "+s"(sums));
    asm volatile("" : "+s"(scratch_rows), "+s"(top_rows));
    asm volatile("" : "+v"(col4));
    double rho_l = rho;
    asm volatile("" : "+v"(rho_l));
    const double inv_nt_s = sgpr_f64(inv_nt), inv_t_s = sgpr_f64(inv_t);
    const ScalarWeight w_all_s = {sgpr_f64(w_all)}, w_mis_s = {sgpr_f64(w_mis)};
    double Bs = 0.0;  // the emission-weighted sum of the beta row (:1046-1063)
    // topology row j = alpha row j * beta row j (the state block), alpha from its checkpoint in HBM
    auto product_from_checkpoint = [&](int j, int c) __attribute__((always_inline)) {
      const GlobalDouble* __restrict__ ar = at_lane(as_global(scratch_rows + (size_t)c * pitch_l), 2u * col4);
      GlobalFloat* __restrict__ tr = at_lane(as_global(top_rows + (size_t)j * pitch_l), col4);
      // in groups of 8 words: all loads of a group, then its products and stores (left alone, the compiler
      // keeps two loads in flight and pays a memory round trip per word)
      constexpr int CH = 8;
      static_for<0, (MPL + CH - 1) / CH>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        constexpr int n = MPL - CH * g < CH ? MPL - CH * g : CH;
        double av[n];
        static_for<0, n>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          av[i] = ar[(size_t)(CH * g + i) * 64];
        });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, n>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          tr[(size_t)(CH * g + i) * 64] = (float)(av[i] * state_get<CH * g + i>());
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    // this lane's place in the parked checkpoint row (after the 2 x G doubles of the block sums)
    uint32_t park_addr = (uint32_t)(uintptr_t)(lds_dyn + 2 * G) + 8u * ((uint32_t)wbase * 64u + (uint32_t)lane);
    asm volatile("" : "+v"(park_addr));
    const double last_coef = sgpr_f64(ka.last_coef[t]), last_nx = sgpr_f64(ka.last_nx[t]);  // (:737-750) the last site's own gap
    const int32_t* dk_s = reinterpret_cast<const int32_t*>(sgpr64((uint64_t)(uintptr_t)(dk + jb)));
    const SiteRec* rec_s = reinterpret_cast<const SiteRec*>(sgpr64((uint64_t)(uintptr_t)(rec + jb)));
    const int nblk = (nds + B - 1) / B;
    for (int c = nblk - 1; c >= 0; c--) {
      const int r0 = c * B;
      const int r1 = r0 + B < nds ? r0 + B : nds;  // rows [r0, r1) of the section
      // The first backward step of the block computes row jtop (below r0: a last block of one row, no step).  What it
      // needs from memory is asked for here, with the loads of the forward run, so that the block pays one round trip
      // to memory for its scalars instead of one per use.
      const int jtop = c == nblk - 1 ? nds - 2 : r1 - 1;
      const int jt = jtop >= r0 ? jtop : r0;  // an address inside the list when no step is taken
      // (every address stays inside the target's lists, also where the value is not used)
      const int i_f1 = jb + r0 + 1 < nd_chunk ? r0 + 1 : nd_chunk - 1 - jb, i_f2 = jb + r0 + 2 < nd_chunk ? r0 + 2 : nd_chunk - 1 - jb;
      const BlockSeeds seeds = load_block_seeds(dk_s + (jb + jt > 0 ? jt - 1 : -jb), dk_s + jt, dk_s + (jt + 1 < nds ? jt + 1 : nds - 1), dk_s + i_f1,
                                                dk_s + i_f2, rec_s + (jt + 1 < nds ? jt + 1 : nds - 1), rec_s + r0, sums + c);
      const int seed_cur = seeds.dk_cur, seed_prev = seeds.dk_prev, seed_n = seeds.dk_next;
      const double seed_coef = jtop == nds - 2 ? last_coef : seeds.coef_top;
      const double seed_nx = jtop == nds - 2 ? last_nx : seeds.nx_top;
      if constexpr (B > 1) {
        // beta waits in its slot while the state block runs the forward recursion over the block again
        if (c != nblk - 1) slot_save<B - 1, MPL>();
        // the checkpoint into the state block (one round trip to memory for the whole row), and a copy of it into LDS
        // for the product of the block's first row, which is the last one the backward sweeps reach
        state_load_row<MPL>(as_global(scratch_rows + (size_t)c * pitch_l), 2u * col4);
        state_loads_wait();
        park_state<MPL>(park_addr);
        if (r1 - r0 > 1) {
          double S = seeds.sum;
          int s_cur = seeds.dk_f1;
          int s_next = seeds.dk_f2;
          double coef_j = seeds.coef_f, nx_j = seeds.nx_f;
          set_site_pointer(rec_s + r0 + 1);
          const uint64_t* rp = rows_w + (size_t)s_cur * stride_l;
          const uint64_t* rp_nx = rows_w + (size_t)s_next * stride_l;
          sload_round<0, RowRounds<MPL>::size(0)>(rp);
          if constexpr (G > 1) {
            set_site_terms(sgpr_f64(coef_j), sgpr_f64(nx_j));
            __syncthreads();  // the entries of the block sums pass from block_sum_hi to exchange_sum
          }
          for (int i = 1; r0 + i < r1; i++) {
            double lg;
            forward_site(r0 + i, S, coef_j, nx_j, rp, rp_nx, s_next, rho_l, stride_l, lg);
            slot_save_at<MPL, B - 1>(i - 1);
          }
          swait();  // the row words asked for beyond the block: the registers go to the backward sweeps next
          if constexpr (G > 1) __syncthreads();  // ... and back
        }
      }
      if (c == nblk - 1) {
        // the last row of the section: beta is the stone (:935-956), its product with alpha is taken as it stands
        double acc = 0.0;
        GlobalFloat* trow = as_global(top_rows + (size_t)(nds - 1) * pitch_l);
        static_for<0, MPL>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          const int n = (wbase + i) * 64 + lane;
          const uint64_t w = row_last[i];
          const bool m = __builtin_amdgcn_inverse_ballot_w64(~w & sk_last);
          const double b = (n < N && n != k) ? (double)beta0[n] : 0.0;
          state_set<i>(b);
          acc += (m ? theta_d : ntheta_d) * b;
        });
        // alpha of that row: the checkpoint itself if the last block has one row, else the slot filled last
        if (B > 1 && r1 - 1 > r0) slot_product_at<MPL, B - 1>(r1 - 2 - r0, trow, col4);
        else product_from_checkpoint(nds - 1, c);
        Bs = block_sum_regs<G>(red, slot, acc);
        if (threadIdx.x == 0) ka.logscales[row0 + nds - 1] = ka.logscales[row0 + nds - 1] + ls_beta;  // float + float (:929)
      } else {
        if constexpr (B > 1) slot_load<B - 1, MPL>();
      }
      if (jtop < r0) continue;  // a last block of one row
      // the backward sweeps of rows jtop .. r0: scalars of the first step fetched here, of the others during the sweeps
      double coef_j = seed_coef, nx_j = seed_nx;
      set_site_pointer(rec_s + jtop);
      int s_cur = seed_cur;
      int s_prev = seed_prev;
      const uint64_t* rp_n = rows_w + (size_t)seed_n * stride_l;
      const uint64_t* rp_c = rows_w + (size_t)s_cur * stride_l;
      sload_row2<kBufA, 0, RowRounds2<MPL>::size(0)>(rp_n);
      sload_row2<kBufC, 0, RowRounds2<MPL>::size(0)>(rp_c);
      for (int j = jtop; j >= r0; j--) {
        {
          if ((j & 31) == 0) rotate_priority(rank, j, 0);
          const double rx = Bs * coef_j;
          const double aa = rx * inv_nt_s;
          const double bb = rx * inv_t_s - aa;
          const double c1 = bb * rho_l;
          write_stateless<MPL, true>(rx, inv_nt_s);
          prefetch_site<false>();
          const uint64_t skn = edge_select(j, nds - 2, sk_last);
          const uint64_t skc = edge_select(j, 0, sk_first);
          prev += nx_j;
          sload_float(ls_rows + j);  // the forward launch's logscale of this row
          double a0, a1, m0 = 0.0, m1 = 0.0;
          const uint64_t* rp_p = rows_w + (size_t)s_prev * stride_l;
          sweep_two_rows<MPL>(a0, a1, m0, m1, aa, rho_l, c1, skn, skc, rp_n, rp_c, rp_p);
          uint32_t B_high;
          Bs = block_sum_hi<G>(red, slot, B_high, a0 + a1, w_all_s, m0 + m1, w_mis_s);
          const float ls_fwd = sloaded_float();
          int s_prev2;
          fetched_site<false>(s_prev2, coef_j, nx_j);
          // topology row j = alpha row j * beta row j, before the beta row is rescaled (:1067-1095)
          if constexpr (B == 1) product_from_checkpoint(j, c);
          else if (j == r0) parked_product<MPL>(park_addr, as_global(top_rows + (size_t)j * pitch_l), col4);
          else slot_product_at<MPL, B - 1>(j - r0 - 1, as_global(top_rows + (size_t)j * pitch_l), col4);
          float lsj = (float)((double)ls_fwd + prev);  // (:994-997)
          bool out_of_band;
          if constexpr (G == 1) out_of_band = out_of_range(Bs);
          else out_of_band = !surely_in_range(B_high) && repaint_out_of_range_exact(Bs);
          if (out_of_band) {
            const double inv = 1.0 / Bs;
            static_for<0, MPL>([&](auto wc) __attribute__((always_inline)) {
              const double f = inv;
              state_scale<decltype(wc)::value>(f);
            });
            const double lg = log(Bs);  // std::log on this side too (:1091-1092)
            prev += lg;
            lsj = (float)((double)lsj + lg);
            Bs = 1.0;
          }
          if (threadIdx.x == 0) as_global(ls_rows)[j] = lsj;
          s_cur = s_prev;
          s_prev = s_prev2;
          rp_n = rp_c;
          rp_c = rp_p;
        }
      }
      swait();  // the row words asked for beyond the block
    }
  }
}

// The two launches as kernels of their own: they differ in the register budget they are compiled with.
template <int MPL, int G>
__global__ void RELATE_REGS_ATTR __launch_bounds__(64 * G) repaint_fwd_kernel_r(const RepaintKArgs ka)
{
  repaint_body<MPL, G, false>(ka);
}
template <int MPL, int G>
__global__ void RELATE_BWD_REPAINT_ATTR __launch_bounds__(64 * G) repaint_bwd_kernel_r(const RepaintKArgs ka)
{
  repaint_body<MPL, G, true>(ka);
}

typedef RepaintLaunch RepaintArgs;  // paint_kernels.h

template <int MPL, int G>
inline hipError_t launch_repaint_t(const RepaintArgs& a, hipStream_t st)
{
  constexpr size_t lds_bytes = (size_t)2 * G * 8;
  RepaintKArgs ka;
  ka.rows = a.rows;
  ka.dk_all = a.dk;
  ka.rec_all = a.rec;
  ka.off = a.off;
  ka.nd_all = a.nd;
  ka.bbj_all = a.bbj;
  ka.bej_all = a.bej;
  ka.order = a.order;
  ka.stones = a.stones;
  ka.row_off = a.row_off;
  ka.last_coef = a.last_coef;
  ka.last_nx = a.last_nx;
  ka.scratch = a.scratch;
  ka.cp_sum = a.cp_sum;
  ka.topology = a.topology;
  ka.logscales = a.logscales;
  ka.w = a.w;
  ka.pitch = a.pitch;
  ka.p.row_stride = a.row_stride;
  ka.p.N = a.N;
  ka.p.L = a.L;
  ka.p.W = a.W;
  ka.p.k_begin = a.k_begin;
  ka.p.K = a.K;
  ka.p.c = a.c;
  if (a.pitch != G * MPL * 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL((repaint_fwd_kernel_r<MPL, G>), dim3((unsigned)a.K), dim3(64 * G), lds_bytes, st, ka);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // the backward launch parks one row of the state block per workgroup behind the block sums
  constexpr size_t lds_bwd = lds_bytes + (repaint_block(MPL) > 1 ? (size_t)G * MPL * 512 : 0);
  static_assert(lds_bwd <= 160 * 1024, "LDS of a CU");
  if (lds_bwd > 48 * 1024) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&repaint_bwd_kernel_r<MPL, G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bwd);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((repaint_bwd_kernel_r<MPL, G>), dim3((unsigned)a.K), dim3(64 * G), lds_bwd, st, ka);
  return hipGetLastError();
}

template <int G>
inline hipError_t launch_repaint_g(const RepaintArgs& a, int MPL, hipStream_t st)
{
  switch (MPL) {
#define RELATE_CASE(M) \
  case M: return launch_repaint_t<M, G>(a, st);
    RELATE_CASE(4) RELATE_CASE(8) RELATE_CASE(12) RELATE_CASE(16) RELATE_CASE(20) RELATE_CASE(24) RELATE_CASE(28)
    RELATE_CASE(32) RELATE_CASE(36) RELATE_CASE(40) RELATE_CASE(48) RELATE_CASE(56) RELATE_CASE(64) RELATE_CASE(72)
    RELATE_CASE(80)
#undef RELATE_CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace regs
}  // namespace relate
