/*
 * relate_gpu.h -- C ABI of librelate_gpu.so: the MI355X (gfx950) implementation
 * of Relate's chromosome painting (stepping-stone Li-Stephens forward-backward).
 *
 * Everything here is plain C: pointers and sizes, no C++ or torch types.  All
 * functions return 0 on success and a negative RELATE_E* code on failure
 * (relate_last_error() gives the text); the library never calls exit() or
 * aborts, unlike the reference (assert / exit(1), src/data.cpp:15-26).
 *
 * Reference citations are relative to the reference tree (tcztzy/relate).
 */
#ifndef RELATE_GPU_H
#define RELATE_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RELATE_OK 0
#define RELATE_EINVAL (-1)  /* bad argument / malformed input */
#define RELATE_EIO (-2)     /* file could not be opened, read or written */
#define RELATE_EGPU (-3)    /* HIP runtime error, or no usable device */
#define RELATE_ENOMEM (-4)  /* host or device allocation failed */
#define RELATE_ERANGE (-5)  /* size outside what the kernels support */

/* Text of the last error on the calling thread ("" if none). */
const char* relate_last_error(void);
/* Number of HIP devices visible (0 when there is none; never fails). */
int relate_device_count(void);
/* Bytes of one stepping-stone record: 56 + 8N (src/fast_painting.cpp:603-615,
 * src/collapsed_matrix.hpp:240-252). */
size_t relate_record_bytes(int N);

/* ------------------------------------------------------------------------
 * 1. Whole-stage drop-in.
 *
 * Replaces  int Paint(cxxopts::ParseResult&, int chunk_index)  (pipeline/Paint.cpp:15-91)
 * and the Rust  Paint::execute  (src/pipelines.rs:133-197), written in the
 * style of the other stages exported through src/ffi.hpp:8-11 (output
 * directory, chunk index, nullable pointers for optional values).
 * Reads <output>/parameters_c<c>.bin and <output>/chunk_<c>.{hap,r}, writes
 * <output>/chunk_<c>/paint/relate_<w>.bin byte-compatible with the reference.
 * theta == NULL keeps Data's default 0.001 (src/data.cpp:97); rho == NULL
 * leaves r unscaled.  device < 0 shards the targets over every visible device
 * (ranges of equal numbers of derived sites; the first device packs the matrix,
 * the others receive the packed copy from it), or over the devices listed in the
 * environment variable RELATE_DEVICES ("0,2,3"; an index may repeat: "0,0" runs
 * two plans with a host thread each on one GPU).  The files appear complete or
 * not at all: records are written to relate_<w>.bin.part and renamed when every
 * window is complete; on failure no relate_<w>.bin of this chunk is left.
 * ------------------------------------------------------------------------ */
int relate_paint_chunk(const char* output_dir, int chunk_index, const double* theta, const double* rho, int device);

/* ------------------------------------------------------------------------
 * 2. File-free batched core on host buffers.
 *
 * Replaces the loop  for k: FastPainting::PaintSteppingStones(data, window_boundaries, pfiles, k)
 * (pipeline/Paint.cpp:78-82, src/fast_painting.cpp:634-637) for targets
 * [k_begin, k_end).  seq is Data::sequence (L x N bytes '0'/'1', row = SNP),
 * r is Data::r already multiplied by rho, theta_painter builds the painter's
 * constants (src/fast_painting.hpp:19-45) and theta_data is Data::theta (the
 * Rust caller passes two different values, src/pipelines.rs:125,181).
 * out receives W x (k_end-k_begin) records, window-major: record (w,k) at
 * ((w*(k_end-k_begin)) + (k-k_begin)) * relate_record_bytes(N), i.e. slab w
 * is the byte range [k_begin*R, k_end*R) of relate_<w>.bin.
 * ------------------------------------------------------------------------ */
int relate_paint_stones(int N, int L, const uint8_t* seq, const double* r, double theta_painter, double theta_data,
                        int num_windows, const int32_t* window_boundaries, int k_begin, int k_end, uint8_t* out,
                        int device);

/* ------------------------------------------------------------------------
 * 3. Object mirror of what the reference's FFI binds (src/lib.rs:4-21):
 *    Data::new2 / Data::SetPainting / FastPainting::new / PaintSteppingStones1.
 * ------------------------------------------------------------------------ */
typedef struct relate_data relate_data;
typedef struct relate_painter relate_painter;

/* Data::Data(6 file names, Ne, mu)  (src/data.cpp:88-99; bound as Data::new2, src/pipelines.rs:161-163).
 * Only the .hap and .r files are read (src/fast_painting.cpp touches nothing else of Data);
 * the other four names may be NULL. */
int relate_data_new(const char* hap, const char* pos, const char* dist, const char* rec, const char* rpos,
                    const char* state, int Ne, double mu, relate_data** out);
/* Data::SetPainting(theta, rho)  (src/data.cpp:124-129). */
int relate_data_set_painting(relate_data* d, double theta, double rho);
int relate_data_dims(const relate_data* d, int* N, int* L);
void relate_data_free(relate_data* d);

/* FastPainting::FastPainting(size_t N, double theta)  (src/fast_painting.hpp:19-31). */
int relate_painter_new(size_t N, double theta, int device, relate_painter** out);
/* void FastPainting::PaintSteppingStones(const Data&, const char* basename, size_t num_windows,
 *                                        const int* window_boundaries, const int k) const
 * (src/fast_painting.cpp:639-652): appends the record of target k to <basename>_<w>.bin for every
 * window.  The caller invokes k = 0..N-1 in order (src/pipelines.rs:179-194); the first call of a
 * run of consecutive k paints a batch of targets on the GPU, later calls are served from that batch,
 * which is kept with the relate_data (the Rust loop builds a new painter per target). */
int relate_painter_paint_stepping_stones(relate_painter* p, const relate_data* d, const char* basename,
                                         size_t num_windows, const int32_t* window_boundaries, int k);
void relate_painter_free(relate_painter* p);

/* ------------------------------------------------------------------------
 * 4. Device-resident plan: inputs and outputs stay in HBM (what the benchmark
 *    and the multi-GPU host drive; same algorithm and kernels as 1-3).
 *    Device pointers are passed as void*; `stream` is a hipStream_t, and NULL
 *    means what it means in every HIP call: the default (null) stream, which is
 *    also the handle a framework such as torch reports for its default stream.
 *    Every call that takes a stream only enqueues work on it (exceptions are
 *    stated); calls on one plan must be issued in order, on one stream or on
 *    streams the caller orders itself.
 * ------------------------------------------------------------------------ */
typedef struct relate_plan relate_plan;

int relate_plan_create(int device, relate_plan** out);
void relate_plan_destroy(relate_plan* p);
/* Kernel choice: variant 1 = register-resident recursion (default), 0 = plain HIP cross-check (N <= 4096);
 * waves_per_target 0 = automatic, else 1/2/4/8.  Call relate_plan_prepare again after changing it. */
int relate_plan_configure(relate_plan* p, int variant, int waves_per_target);

/* Packs Data::sequence ('0'/'1' bytes, device pointer d_seq, L x N) into bit rows and their transpose
 * in HBM ("bit-planes"); replaces ReadSequenceFromBin's in-memory layout (src/data.cpp:541-550).
 * d_r: L doubles on the device (Data::r times rho).  wb: host array of num_windows+1 ints.
 * d_seq is read by this call only; d_r is borrowed: relate_plan_prepare and relate_plan_repaint read it again, so
 * the caller keeps it alive (and unchanged) until the plan loads another chunk or is destroyed. */
int relate_plan_load_device(relate_plan* p, int N, int L, const void* d_seq, const void* d_r, int num_windows,
                            const int32_t* window_boundaries, void* stream);
/* Same from host buffers (copies them to the device first). */
int relate_plan_load_host(relate_plan* p, int N, int L, const uint8_t* seq, const double* r, int num_windows,
                          const int32_t* window_boundaries);
/* Bytes of the packed matrix (bit rows, then their transpose) of an N x L chunk. */
size_t relate_packed_bytes(int N, int L);
/* As relate_plan_load_device, but packs into the caller's device buffer d_packed (relate_packed_bytes(N, L)
 * bytes, 256-byte aligned) and keeps using it: the caller owns it and must keep it alive while the plan
 * uses the chunk.  This is the buffer a host broadcasts to the other devices (RCCL) before they call
 * relate_plan_load_packed. */
int relate_plan_load_device_into(relate_plan* p, int N, int L, const void* d_seq, const void* d_r, int num_windows,
                                 const int32_t* window_boundaries, void* d_packed, size_t packed_bytes, void* stream);
/* The packed matrix (rows then transpose, one allocation) for a broadcast to other devices. */
int relate_plan_packed(relate_plan* p, void** d_packed, size_t* bytes);
/* Adopts an already packed matrix (as returned by relate_plan_packed on another device, after a broadcast). */
int relate_plan_load_packed(relate_plan* p, int N, int L, const void* d_packed, size_t bytes, const void* d_r,
                            int num_windows, const int32_t* window_boundaries, void* stream);

/* Per-target precompute for [k_begin,k_end): derived-site lists, per-gap recombination terms and
 * window bookkeeping (src/fast_painting.cpp:33-170).  Returns the sum of nd_k in *sum_derived.  The call waits for
 * the counts (it sizes its buffers with them) but not for the kernels it launches after that: like
 * relate_plan_paint it is asynchronous with respect to `stream`, and a later call on the same stream follows
 * in order.  relate_plan_repaint_layout uses a geometry of its own (see relate_plan_geometry): its pitch may
 * differ from 64 * waves_per_target * states_per_lane of the painting. */
int relate_plan_prepare(relate_plan* p, double theta_painter, double theta_data, int k_begin, int k_end,
                        int64_t* sum_derived, void* stream);
/* Forward + backward recursion of the prepared targets (src/fast_painting.cpp:214-599); writes
 * num_windows x K records into d_out (device pointer, relate_record_bytes(N)*K*num_windows bytes). */
int relate_plan_paint(relate_plan* p, void* d_out, void* stream);
/* Re-painting of one window from its stepping stones for the prepared targets:
 *   void FastPainting::RePaintSection(const Data&, CollapsedMatrix<float>& topology, std::vector<float>& logscales,
 *        CollapsedMatrix<float>& alpha_begin, CollapsedMatrix<float>& beta_end, int boundarySNP_begin,
 *        int boundarySNP_end, float logscale_alpha, float logscale_beta, const int k) const
 * (src/fast_painting.cpp:654-1126), called for k = 0..N-1 by DistanceMeasure::GetTopologyWithRepaint
 * (src/anc_builder.cpp:53-110).  d_stones: the records of window `window` of the prepared targets as
 * relate_plan_paint wrote them (slab `window` of its output, K x relate_record_bytes(N) bytes, device
 * pointer).  Target t gets rows row_offsets[t] .. row_offsets[t+1]-1 (one per site of its section: the stone
 * sites and every site in between at which it carries the derived allele); row i of d_topology starts at
 * float index i*pitch and holds N floats alpha*beta (pitch >= N, the rest of the row is zero), d_logscales[i]
 * is its logscale.  The painter's theta is the one given to relate_plan_prepare as theta_painter; the
 * reference builds this painter from Data, i.e. with theta_painter == theta_data.
 * relate_plan_repaint_layout only computes the layout (row_offsets: K+1 entries, may be NULL). */
int relate_plan_repaint_layout(relate_plan* p, int window, int64_t* total_rows, int* pitch, int64_t* row_offsets);
int relate_plan_repaint(relate_plan* p, int window, const void* d_stones, void* d_topology, void* d_logscales,
                        void* stream);
/* The distance matrix of one SNP from a device-resident topology:
 *   void DistanceMeasure::GetMatrix(const int snp)  (src/anc_builder.cpp:113-211), the part after the
 * topology has been loaded (:120-196).  The caller keeps the reference's bookkeeping (v_snp_prev, v_rpos_prev,
 * v_rpos_next, src/anc_builder.cpp:85-105,:144-161) and passes, for every row n, host arrays:
 *   row_index[n]   = row_offsets[n] + v_snp_prev[n] (row of relate_plan_repaint's topology),
 *   interpolate[n] = 0 where the reference takes that row as is (n carries the derived allele at snp, or snp
 *                    is the first or last SNP), 1 where it mixes rows row_index[n] and row_index[n]+1,
 *   weight_left[n], weight_right[n] = the two weights of that mix (:152-161; ignored where interpolate[n] == 0).
 * d_matrix receives N x N floats, row n = distances from n (matrix[n][m] of the reference), row minimum
 * subtracted, zero diagonal.  total_rows bounds the row indices.
 * The two exponentials of a mixed row, exp(logscale_prev - logscale_next) and its inverse (:171-172), are taken
 * on the host with the C library's exp, as the reference takes them, so that the matrix is the reference's bit
 * for bit (its consumer compares distances with ==, src/tree_builder.cpp:279-299): the call gathers the 2N
 * logscales it needs on the device, fetches them -- which waits for `stream` once -- and then enqueues the
 * kernel.  relate_distance_get_matrix below keeps a host copy of the section's logscales and never waits. */
int relate_plan_distance_matrix(relate_plan* p, int N, const void* d_topology, int pitch, const void* d_logscales,
                                int64_t total_rows, const int64_t* row_index, const uint8_t* interpolate,
                                const double* weight_left, const double* weight_right, void* d_matrix, void* stream);
/* relate_plan_distance_matrix for callers that cannot allocate: loops that ask for one matrix per tree inside a
 * serving process, harnesses that forbid hipMalloc.  It has no counterpart in the reference (whose GetMatrix
 * fills a host matrix in place); same arguments, same kernel and same result as the entry point above, without
 * a plan.  The per-row arguments (41 bytes per row) are staged in a fixed array inside the library's code
 * object, of which every GPU holds its own copy (1.44 MiB) until the process ends, and in a fixed host array of
 * the same size: N may be at most 36 864, larger values are refused with RELATE_ERANGE.  The call allocates
 * nothing on the device or the host; it copies the four host arrays (pageable memory is fine), waits for
 * `stream` (the logscales come back for the exponentials, and the host array must be free again on return) and
 * returns once the matrix has been computed.  It is therefore not a call to make under HIP graph capture.
 * All calls on one GPU share that GPU's array, and all calls of the process the host array: the library holds a
 * lock for the whole call, so concurrent callers are served one after the other instead of racing. */
int relate_distance_matrix_noalloc(int device, int N, const void* d_topology, int pitch, const void* d_logscales,
                                   int64_t total_rows, const int64_t* row_index, const uint8_t* interpolate,
                                   const double* weight_left, const double* weight_right, void* d_matrix, void* stream);
/* DistanceMeasure with the topology of the current section and the matrix in HBM: the object
 * AncesTreeBuilder::BuildTopology (src/anc_builder.cpp:265-421) asks for the distance matrix of a SNP,
 *   class DistanceMeasure { ...; void GetTopologyWithRepaint(const int snp); void GetMatrix(const int snp); }
 * (src/anc_builder.hpp:35-90, src/anc_builder.cpp:53-211).  The reference's bookkeeping (section,
 * section_startpos, section_endpos, v_snp_prev, v_rpos_prev, v_rpos_next) is kept on the host under its own
 * names; re-painting and the matrix run on the device through the two plan calls above.
 * plan: chunk loaded, prepared for all N targets with theta_painter == theta_data, painted; d_stones: what
 * relate_plan_paint wrote (device pointer); seq: Data::sequence (L x N bytes '0'/'1'), rpos: Data::rpos
 * (L + 1 doubles), both host, both copied.  first_section: the window the first call re-paints (0 for a
 * chunk walked from its start).  The plan and d_stones must outlive the object. */
typedef struct relate_distance relate_distance;
int relate_distance_new(relate_plan* plan, const uint8_t* seq, const double* rpos, const void* d_stones, int first_section,
                        relate_distance** out);
void relate_distance_free(relate_distance* d);
/* GetTopologyWithRepaint(snp): re-paints the next section for every target and resets the bookkeeping for
 * `snp`, which must lie inside that section (the reference asserts it, :106-107). */
int relate_distance_get_topology_with_repaint(relate_distance* d, int snp, void* stream);
/* What BuildTopology does to the bookkeeping when it moves on to `snp` (src/anc_builder.cpp:331-335). */
int relate_distance_advance(relate_distance* d, int snp);
/* GetMatrix(snp): re-paints first if snp lies beyond the current section (:115-118); N x N floats into d_matrix
 * (device pointer), row minimum subtracted, zero diagonal.  Enqueues on `stream`. */
int relate_distance_get_matrix(relate_distance* d, int snp, void* d_matrix, void* stream);
/* The bookkeeping as it stands (any pointer may be NULL; the vectors have N entries). */
int relate_distance_state(const relate_distance* d, int* section, int* section_startpos, int* section_endpos,
                          int64_t* v_snp_prev, double* v_rpos_prev, double* v_rpos_next);
/* ------------------------------------------------------------------------
 * The tree builder's opening scans over the distance matrix:
 *   void MinMatch::Initialize(CollapsedMatrix<float>& d, std::uniform_real_distribution<double>& dist_unif)
 *   (src/tree_builder.cpp:57-127; with sample ages :130-214; declared src/tree_builder.hpp:71-72), called by
 *   MinMatch::QuickBuild (:1003, :928) on the matrix DistanceMeasure::GetMatrix has just filled.
 * Its two passes over the N x N matrix run on the device, on the matrix where relate_distance_get_matrix /
 * relate_plan_distance_matrix left it:
 *   min_values[i] = min over l != i of d[i][l], + threshold                               (:64-78)
 *   the pairs i < j with min_values[i] >= d[i][j] and min_values[j] >= d[j][i]            (:81-91)
 * and the pairs come back compacted, in the order of the reference's double loop (i ascending, then j
 * ascending), each with sym_dist = d[i][j] + d[j][i] (:95).  What stays on the host is the part that is
 * sequential by construction: one draw of the caller's generator per pair, in that order (rng.seed(1), :852),
 * and the bookkeeping of mcandidates / best_candidate (:96-114), which relate_minmatch_apply_candidates does
 * with the caller's generator behind a callback.  The coalescence loop itself (Coalesce, :258-475) is not part
 * of this library.
 * threshold: MinMatch's, -0.2 * log(theta / (1 - theta)) as a float (:43) = relate_minmatch_threshold(theta).
 * min_values: N floats (host).  cand_i / cand_j / cand_dist: `capacity` entries each (host); *count receives
 * the number of pairs.  If there are more pairs than `capacity`, nothing is stored, *count still holds the
 * number and the call returns RELATE_ERANGE: call again with larger arrays.  The call waits for `stream`.
 * ------------------------------------------------------------------------ */
float relate_minmatch_threshold(double theta);
int relate_plan_minmatch_scan(relate_plan* p, int N, const void* d_matrix, float threshold, float* min_values,
                              int64_t capacity, int32_t* cand_i, int32_t* cand_j, float* cand_dist, int64_t* count,
                              void* stream);
/* struct Candidate (src/tree_builder.hpp:12-25) as plain data. */
typedef struct relate_candidate {
  int32_t lin1, lin2;
  double dist, dist2, dist3;
  int32_t replace;
} relate_candidate;
/* The rest of Initialize on the host: for every pair of the list, in order, one call of draw(draw_ctx) (the
 * caller's dist_unif(rng)) and the updates of mcandidates[i], mcandidates[j] and best_candidate
 * (:96-114; sample_ages != NULL: :169-200 with the caller's current `age`, :926).  mcandidates: N entries, whose
 * dist / dist2 (and dist3 / replace with sample ages) are reset first, as :66-67 / :139-142 do; best: the
 * caller's best_candidate, updated in place (QuickBuild sets its distances to infinity before, :889-891). */
int relate_minmatch_apply_candidates(int N, int64_t count, const int32_t* cand_i, const int32_t* cand_j,
                                     const float* cand_dist, const double* sample_ages, double age,
                                     double (*draw)(void*), void* draw_ctx, relate_candidate* mcandidates,
                                     relate_candidate* best);

/* nd_k of every haplotype of the loaded chunk (the number of sites its list holds: its derived sites and the
 * two forced ends, src/fast_painting.cpp:60-110), N ints on the host; the cost of painting target k is
 * proportional to it.  Waits for the result. */
int relate_plan_derived_counts(relate_plan* p, int32_t* nd);
/* Contiguous partition of N targets into `parts` ranges of (nearly) equal sums of nd: bounds[i] .. bounds[i+1]
 * is range i, bounds has parts + 1 entries, bounds[0] = 0, bounds[parts] = N.  Host-only. */
int relate_shard_by_work(int N, const int32_t* nd, int parts, int32_t* bounds);
/* Paints targets [k_begin, k_end) of the loaded chunk and writes record k of window w at byte k *
 * relate_record_bytes(N) of the open file fds[w] (pwrite; num_fds must equal the number of windows), i.e. the
 * loop of pipeline/Paint.cpp:78-82 for a shard of the targets, whatever the order in which shards finish.
 * Batches the targets to the free device memory and copies the records back through page-locked staging
 * while host threads write them.  Returns when everything has been written. */
int relate_plan_paint_to_files(relate_plan* p, double theta_painter, double theta_data, int k_begin, int k_end,
                               const int* fds, int num_fds);
/* Frees the page-locked staging block (at most 1 GiB) that the host-buffer and file entry points keep between
 * calls; a no-op while a call is using it. */
void relate_release_host_staging(void);
/* Milliseconds of the last pack / prepare / paint, from HIP events on the stream they ran on
 * (blocks until the paint has finished); after relate_plan_repaint, ms_paint is the time of its two launches. */
int relate_plan_timings(relate_plan* p, float* ms_pack, float* ms_prepare, float* ms_paint);
/* Geometry the recursion kernel was launched with.  relate_plan_prepare chooses it from the panel size and the
 * number of targets of the range (how many rounds of waves the launch makes over the chip), so two ranges of one
 * chunk may be painted with different geometries; the records do not depend on it. */
int relate_plan_geometry(const relate_plan* p, int* waves_per_target, int* states_per_lane);
/* Geometry of the re-painting launches of the prepared range (its own: relate_plan_repaint_layout's pitch is
 * 64 * waves_per_target * states_per_lane of THIS call) and the number of rows between two alpha checkpoints: the
 * launches move 4 + 16 / checkpoint_rows bytes per topology element (1: every alpha row is kept, as
 * FastPainting::RePaintSection keeps them, src/fast_painting.cpp:786-919). */
int relate_plan_repaint_geometry(const relate_plan* p, int* waves_per_target, int* states_per_lane, int* checkpoint_rows);

/* GPU fast_log (src/fast_log.hpp:6-21) on n floats, host in / host out: used by the parity tests. */
int relate_fast_log(const float* in, float* out, int n, int device);

/* ------------------------------------------------------------------------
 * 5. Synthetic chunks (seeded; no counterpart in the reference, which has no
 *    simulator): haplotypes with linkage disequilibrium from a tree that
 *    changes by prune-and-regraft moves along the chunk, a 1/i site-frequency
 *    spectrum, and a recombination map that is flat or has hotspots.
 *    seq: L*N bytes, r: L doubles, both host buffers supplied by the caller.
 * ------------------------------------------------------------------------ */
int relate_synth_chunk(int N, int L, uint64_t seed, int skewed_map, uint8_t* seq, double* r);
/* Writes <output>/parameters_c<c>.bin and <output>/chunk_<c>.{hap,bp,dist,r,rpos,state} in the
 * formats of Data::MakeChunks (src/data.cpp:272-286,:496-526). */
int relate_write_chunk(const char* output_dir, int chunk_index, int N, int L, const uint8_t* seq, const double* r,
                       int num_windows, const int32_t* window_boundaries);
/* Window boundaries by the reference's memory rule (src/data.cpp:235-245): a window closes when
 * sum over its SNPs of derived*(N+1) floats reaches the budget.  wb must hold max_windows+1 ints;
 * returns the number of windows (>0) or a negative code. */
int relate_window_boundaries(int N, int L, const uint8_t* seq, double memory_gb, int max_windows, int32_t* wb);

/* ------------------------------------------------------------------------
 * 6. Phased haplotypes straight into bit rows: the input side of
 *      void Data::MakeChunks(const std::string& filename_haps, const std::string& filename_sample,
 *                            const std::string& filename_map, const std::string& filename_dist,
 *                            const std::string& file_out, bool use_transitions, float min_memory)
 *    (src/data.cpp:134-528) without its byte-per-allele matrices: every SNP is packed into N bits as it is
 *    parsed (haps::ReadSNP, src/data.cpp:553-580), and a chunk goes to the device as bit rows (an eighth of
 *    chunk_<c>.hap).  Chunks, their overlap of 20 000 SNPs and their windows follow the reference's rules
 *    (src/data.cpp:145-267), r and rpos its map interpolation (:452-491; filename_dist "unspecified": :397-415).
 *    .haps, .sample and the map may be gzip-compressed (read through `gunzip -c`, as gzip::open does).
 *    Where the reference asserts or exits (unsorted positions, a line with too few alleles, too small a memory
 *    allowance) these calls return an error code.
 * ------------------------------------------------------------------------ */
typedef struct relate_haps relate_haps;
int relate_haps_open(const char* haps_path, const char* sample_path, const char* map_path, double memory_gb,
                     relate_haps** out);
void relate_haps_close(relate_haps* h);
int relate_haps_dims(const relate_haps* h, int* N, int* L, int* num_chunks);
/* First SNP of chunk `chunk` in the data set, its number of SNPs (overlap included) and of windows: the
 * contents of parameters_c<chunk>.bin, whose boundaries (num_windows + 1 ints, relative to first_snp)
 * relate_haps_chunk_windows copies out. */
int relate_haps_chunk_dims(const relate_haps* h, int chunk, int* first_snp, int* L_chunk, int* num_windows);
int relate_haps_chunk_windows(const relate_haps* h, int chunk, int32_t* wb);
/* Whole-data-set vectors, any of which may be NULL: bp (L + 1 ints, the last one base past the last SNP),
 * dist (L), rpos (L + 1 doubles, Morgans), r (L doubles: 2500 * max(rpos[s+1] - rpos[s], 1e-10)). */
int relate_haps_positions(const relate_haps* h, int32_t* bp, int32_t* dist, double* rpos, double* r);
/* The bit matrix on the host: row s starts at rows + s * row_stride_words; bit n of a row (word n / 64, bit
 * n % 64) is set where haplotype n carries the derived allele, and every bit from N on is set.  The pointer
 * stays valid until relate_haps_close. */
int relate_haps_rows(const relate_haps* h, const uint64_t** rows, int* row_stride_words);
/* Words per bit row for N haplotypes (the layout of relate_haps_rows and relate_plan_load_rows_host). */
int relate_row_stride_words(int N);
/* Writes what MakeChunks writes into output_dir, from the bit rows: parameters.bin, props.bin and, per chunk,
 * parameters_c<c>.bin and chunk_<c>.{hap,bp,dist,r,rpos,state} (src/data.cpp:270-368,:380-391,:437-449,:496-526),
 * for the stages of the reference that read them.  use_transitions as in MakeChunks (0: transitions get state 0). */
int relate_haps_write_chunks(const relate_haps* h, const char* output_dir, int use_transitions);
/* Loads chunk `chunk` into a plan as relate_plan_load_host would load chunk_<chunk>.hap and chunk_<chunk>.r:
 * uploads the bit rows and r * rho (Data::SetPainting, src/data.cpp:124-129), builds the transposed matrix
 * on the device.  Returns when the upload has finished; relate_plan_prepare / relate_plan_paint follow as usual. */
int relate_plan_load_haps_chunk(relate_plan* p, const relate_haps* h, int chunk, double rho);
/* The same from a caller's own bit rows (layout of relate_haps_rows; L * relate_row_stride_words(N) words, host
 * memory) and recombination distances. */
int relate_plan_load_rows_host(relate_plan* p, int N, int L, const uint64_t* rows, const double* r, double rho,
                               int num_windows, const int32_t* window_boundaries);

#ifdef __cplusplus
}
#endif
#endif /* RELATE_GPU_H */
