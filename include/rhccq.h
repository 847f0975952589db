/* rhccq.h -- C ABI of the MI355X-native RHCCQ encoder hot path (librhccq_hip.so).
 *
 * Boundary: the reference has no FFI; its "interface" for this path is the Python
 * import surface encoder.compression.{clustering,merging,subregions,regions,image}.  The host side
 * stays in Python (roibasedimagecompression_amd/, mirrored under encoder/ and decoder/) and calls
 * the entry points below through ctypes.  Each entry point is a thin launcher of hand-written
 * gfx950 kernels; the reference function it replaces is cited next to it (paths relative to the
 * reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP, same device as the context) unless the parameter
 *     name ends in _host; buffers are caller-allocated (the Python side uses torch tensors);
 *   - every call is asynchronous on the context's HIP stream unless documented "synchronises";
 *   - return value: 0 on success, negative on error (RHCCQ_E_*); rhccq_last_error() gives text;
 *   - no exceptions cross the boundary; a context is not thread-safe, use one per thread;
 *   - colours travel as "keys": uint32 = R<<16 | G<<8 | B  (lexicographic order == numeric order).
 */
#ifndef RHCCQ_H
#define RHCCQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RHCCQ_OK 0
#define RHCCQ_E_ARG (-1)     /* bad argument / shape */
#define RHCCQ_E_HIP (-2)     /* HIP runtime error */
#define RHCCQ_E_LIMIT (-3)   /* size beyond what the kernel supports */

#define RHCCQ_BITMAP_WORDS 524288u /* 2^24 colours / 32 bits: one unique-colour bitmap per job */
#define RHCCQ_EPS_LDS_MAX 10240    /* eps-components: problems up to this size run from LDS */
#define RHCCQ_KM_LDS_MAX 10240     /* k-means: problems up to this size keep their points in LDS */

typedef struct rhccq_ctx rhccq_ctx;

/* ---- context ------------------------------------------------------------------------------ */
int rhccq_ctx_create(int device, void* hip_stream /* NULL = the default (null) stream */, rhccq_ctx** out);
void rhccq_ctx_destroy(rhccq_ctx* ctx);
const char* rhccq_last_error(const rhccq_ctx* ctx);
/* Tuning knobs.  They only move the thresholds at which a kernel switches between two implementations of the
 * same computation (results never change); the test-suite lowers them to drive the large-input paths with inputs
 * the oracle can still check.
 *   RHCCQ_OPT_INIT_LDS_BLOCKS  k-means++ (rhccq_mbk_init): block tables live in LDS while the init sample has at
 *                              most this many 64-sample blocks (0..4096, default 4096), in global memory beyond;
 *   RHCCQ_OPT_INIT_MAX_ITEMS   capacity of the shared (candidate, block) work list (1..12288, default 12288);
 *                              picks that exceed it evaluate each candidate by its own enumeration instead;
 *   RHCCQ_OPT_INIT_KERNEL      0 (default): the newest k-means++ chain whose tables fit LDS -- third generation (leaves of 16
 *                              samples, box pruning) up to 98 304 init samples, second generation (blocks of 64) up to 262 144,
 *                              first generation beyond; 1: the first-generation chain always; 2: the second generation whenever
 *                              its tables fit; 3: same as 0; 4: brute force with the samples in registers (kpp_flat.h, at most
 *                              8 192 init samples; the chain KMeans uses); 5: the third generation with every candidate's improvement
 *                              summed by the wave that found it (no shared work list, two barriers per pick instead of three: measured
 *                              5 % SLOWER).  Same picks everywhere; the other chains serve other sizes and as
 *                              cross-checks;
 *   RHCCQ_OPT_INIT_CANDS_PER_WAVE  third-generation chain: how many of a pick's candidates ONE search wave finds and descends for,
 *                              as interleaved dependency chains of one instruction stream (1, 2 or 3; same picks);
 *   RHCCQ_OPT_REASSIGN_LDS     mini-batch steps that reassign low-count centres: 1 (default) = the five sweeps over the weights read
 *                              a 32-bit copy in LDS when k <= 30 720, 0 = always global memory (the path of larger k; same results);
 *   RHCCQ_OPT_INIT_SHARDS      workgroups (CUs) per problem of the second-generation chain: 1 (default) = one; 2 / 4 / 8 =
 *                              up to that many, each owning a range of the draws, when every shard keeps >= 4096 init
 *                              samples and at most 64 workgroups result (they wait for each other inside the launch, so
 *                              all must be resident).  Same picks; measured SLOWER than one workgroup (two cross-CU
 *                              exchanges per pick), kept as a checked alternative. */
#define RHCCQ_OPT_INIT_LDS_BLOCKS 1
#define RHCCQ_OPT_INIT_MAX_ITEMS 2
#define RHCCQ_OPT_INIT_KERNEL 3
#define RHCCQ_OPT_INIT_SHARDS 4
#define RHCCQ_OPT_INIT_CANDS_PER_WAVE 5
#define RHCCQ_OPT_REASSIGN_LDS 6
/*   RHCCQ_OPT_REASSIGN_ORDER   THE ONE OPTION THAT CHANGES RESULTS.  Which tied low-count centres a capped reassignment of a mini-batch
 *                              step takes (sklearn _mini_batch_step: np.argsort(weight_sums)[:batch / 2], an unstable sort):
 *                              1 (default) = the slots numpy's scalar aquicksort_<double> fills (csrc/k8_npysort.h; the whole fit then
 *                              equals scikit-learn's untouched fit_predict under NPY_DISABLE_CPU_FEATURES = <AVX512 family> AVX2
 *                              FMA3, the host setting of record), 0 = the stable order (weight, index) of rounds 1-3 (= sklearn
 *                              with that call forced to kind='stable'). */
#define RHCCQ_OPT_REASSIGN_ORDER 7
int rhccq_ctx_set_int(rhccq_ctx* ctx, int32_t option, int64_t value);
int rhccq_sync(rhccq_ctx* ctx);                 /* hipStreamSynchronize on the context stream */
void* rhccq_stream(rhccq_ctx* ctx);             /* the hipStream_t in use */
/* Launch the following entry points on another HIP stream (not owned).  The Python side calls this whenever torch's
 * current stream differs from the bound one, so that the caller's allocations / memsets (torch, current stream) and the
 * kernels stay ordered on ONE stream -- e.g. inside `with torch.cuda.stream(s):`.  Work already queued on the previous
 * stream is not waited for: ordering across streams is the caller's business, as with torch itself. */
int rhccq_ctx_set_stream(rhccq_ctx* ctx, void* hip_stream);
int rhccq_abi_version(void);

/* ---- parameters: compute_clustering_params (encoder/compression/clustering.py:108-135) ----- */
int rhccq_params(int64_t n_colors, double quality, double* eps_host, int64_t* max_colors_host);
/* integer form of the eps predicate: thr, boundary (-1 if none), r2 = (eps/255)^2 */
int rhccq_eps_threshold(double eps, int32_t* thr_host, int32_t* boundary_host, double* r2_host);

/* ---- K0/K1: per-job unique colours ----------------------------------------------------------
 * A "job" is one set of pixels whose unique colours are wanted: a whole crop
 * (get_all_unique_colors, clustering.py:4-103) or one SLIC segment of a region
 * (subregion_quantization, subregions.py:315-426).  Each job owns a 2^24-bit bitmap.
 *
 * rhccq_job_scan: one pass over the pixels.  For class c (0..n_class-1) pixel p belongs to job
 *   job_base[c] + labels[c][p] - 1 when labels[c][p] > 0 (labels[c] == NULL: every pixel belongs
 *   to job job_base[c]).  Sets the colour bit of every non-black pixel, accumulates per-job
 *   stats {min_r, max_r, min_c, max_c, count, n_black} (int32[6] each, caller-initialised to
 *   {INT_MAX,-1,INT_MAX,-1,0,0}); black pixels set their bit only when black_is_colour != 0. */
int rhccq_job_scan(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class,
                   const int32_t* const* labels_host /* host array of device ptrs */,
                   const int32_t* job_base_host, int32_t black_is_colour, uint32_t* bitmaps,
                   int32_t* stats);
/* Same pass with BYTE colour flags (bytemaps: uint8[n_jobs][2^24], zero-initialised): plain idempotent
 * stores instead of scattered device-scope atomics (which run at the memory side on MI355X).
 * rhccq_bytemap_pack then ORs the flags into the 2 MiB bitmaps (32 bytes -> one word). */
int rhccq_job_scan_bytes(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class,
                         const int32_t* const* labels_host, const int32_t* job_base_host,
                         int32_t black_is_colour, uint8_t* bytemaps, int32_t* stats);
int rhccq_bytemap_pack(rhccq_ctx* ctx, const uint8_t* bytemaps, int32_t n_jobs, uint32_t* bitmaps);
/* set bit 0 (black) of the bitmaps of the jobs listed (device int32 list) */
int rhccq_job_set_black(rhccq_ctx* ctx, uint32_t* bitmaps, const int32_t* jobs, int32_t n_jobs);
/* popcount of every job's bitmap -> counts[n_jobs] (device) and chunk sums workspace
 * chunk_sums[n_jobs*512] */
int rhccq_bitmap_count(rhccq_ctx* ctx, const uint32_t* bitmaps, int32_t n_jobs, uint32_t* chunk_sums,
                       int32_t* counts);
/* word_prefix[n_jobs*BITMAP_WORDS][2] = (bitmap word, exclusive prefix of the set bits before it) pairs -- the table the
 * per-pixel passes gather from, one 8-byte load per rank lookup -- and the sorted palette keys of each
 * job written at keys_out[pal_off[j] ...) (np.unique order, clustering.py:22) */
int rhccq_bitmap_emit(rhccq_ctx* ctx, const uint32_t* bitmaps, int32_t n_jobs, const uint32_t* chunk_sums,
                      const int64_t* pal_off /* device int64[n_jobs] */, uint32_t* word_prefix,
                      uint32_t* keys_out);
/* black-in-segment fix (subregions.py:393-421): per job the in-mask non-black pixel with the
 * smallest R^2+G^2+B^2, first in raster order; best[job] = (norm2<<40 | pixel index), init ~0 */
int rhccq_job_blackfix(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class,
                       const int32_t* const* labels_host, const int32_t* job_base_host,
                       const uint8_t* job_needs_fix /* device u8[n_jobs] */, unsigned long long* best);
/* ---- frames with very many segments: unique colours by one device sort instead of 6 MiB of bitmap tables per job.
 * rhccq_job_stats: the statistics half of rhccq_job_scan alone (stats as there).
 * rhccq_job_sort_unique: every masked pixel of every class becomes the key (job << 24 | colour) -- in-mask black recoloured by fix_key[job] when
 *   that is non-zero, and one synthetic (job, colour 0) entry per job listed in black_jobs (crops that show background, all-black segments) --;
 *   one radix sort + head flags + an exclusive scan give, per job in job order, its sorted distinct colours (np.unique order) in
 *   keys_out[job_start[job] ...) (job_start[job] = -1: the job has no colour; *n_unique = their total) and, per (class, pixel), the rank of the
 *   pixel's colour inside its job's palette in rankmap (int32[n_class][H*W], -1 where the pixel has no job).  tmp: rhccq_job_sort_unique_bytes
 *   (n_class * H * W + n_black) bytes of device scratch; keys_out must hold that many entries.
 * rhccq_job_index_ranked / rhccq_frame_remap_ranked: rhccq_job_index (first positions only) / rhccq_frame_remap reading the stored ranks. */
int rhccq_job_stats(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class, const int32_t* const* labels_host,
                    const int32_t* job_base_host, int32_t* stats);
int64_t rhccq_job_sort_unique_bytes(int64_t n_entries);
int rhccq_job_sort_unique(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class, const int32_t* const* labels_host,
                          const int32_t* job_base_host, int32_t n_jobs, const uint32_t* fix_key /* device, may be NULL */,
                          const int32_t* black_jobs /* device int32[n_black] */, int32_t n_black, void* tmp, int64_t tmp_bytes, int32_t* rankmap,
                          uint32_t* keys_out, int32_t* job_start /* device int32[n_jobs] */, int32_t* n_unique /* device int32 */);
int rhccq_job_index_ranked(rhccq_ctx* ctx, int32_t H, int32_t W, int32_t n_class, const int32_t* const* labels_host, const int32_t* job_base_host,
                           const int32_t* rankmap, const int64_t* pal_off, int32_t* first_pos, const int32_t* fp_lut);
int rhccq_frame_remap_ranked(rhccq_ctx* ctx, int32_t H, int32_t W, int32_t n_class, const int32_t* const* labels_host, const int32_t* job_base_host,
                             const int32_t* rankmap, const int64_t* pal_off, const int32_t* lut, const int32_t* lut2, int32_t default_index, void* out,
                             int32_t out_elem_bytes);
/* per-pixel palette index (rank of the pixel's colour in its job's palette) and/or first raster
 * position of every palette entry.  fix_key[job] (device, may be NULL) = key that replaces in-mask
 * black pixels (0 = no fix).  idx_out (int32[n_class][H*W], may be NULL): -1 where the pixel has no
 * job.  first_pos (may be NULL) must be initialised to INT_MAX; entry = pal_off[job]+rank, or
 * fp_lut[pal_off[job]+rank] when fp_lut != NULL (first positions of a clustered, smaller palette). */
int rhccq_job_index(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class,
                    const int32_t* const* labels_host, const int32_t* job_base_host,
                    const uint32_t* bitmaps, const uint32_t* word_prefix, const int64_t* pal_off,
                    const uint32_t* fix_key, int32_t* idx_out, int32_t* first_pos, const int32_t* fp_lut);
/* rhccq_job_index with first positions AND, per (class, pixel), the table entry the pixel shows (fp_lut[pal_off[job] + rank], or
 * pal_off[job] + rank without fp_lut; -1 where the pixel has no job) in entries_out (int32[n_class][H*W]): what the final remap needs,
 * so that it reads neither the pixels nor the rank tables again.  rhccq_frame_remap_entries: the index map from those entries --
 * for every pixel the first class (in the order given) whose label is > 0 decides: out = lut2[entry] (lut2 may be NULL: the entry
 * itself), default_index where no class covers the pixel (clustering.py:373-377 composed with merging.py:64-82). */
int rhccq_job_index_entries(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class,
                            const int32_t* const* labels_host, const int32_t* job_base_host, const uint32_t* bitmaps,
                            const uint32_t* word_prefix, const int64_t* pal_off, const uint32_t* fix_key, int32_t* first_pos,
                            const int32_t* fp_lut, int32_t* entries_out);
int rhccq_frame_remap_entries(rhccq_ctx* ctx, int32_t H, int32_t W, int32_t n_class, const int32_t* const* labels_host,
                              const int32_t* job_base_host, const int32_t* entries, const int32_t* lut2, int32_t default_index,
                              void* out, int32_t out_elem_bytes);

/* ---- K3/K4: DBSCAN(min_samples=1) labels = eps-graph components (clustering.py:233-235) -------
 * problems p = 0..n_prob-1: keys[off[p] .. off[p]+n[p]); labels in sklearn order (rank of the
 * component's smallest member index).  desc: int32[n_prob][4] = {off, n, thr, boundary},
 * r2: double[n_prob].  labels_out int32 (same offsets), ncomp_out int32[n_prob]. */
int rhccq_eps_components(rhccq_ctx* ctx, const uint32_t* keys, const int32_t* desc, const double* r2,
                         int32_t n_prob, int32_t max_n, int32_t* labels_out, int32_t* ncomp_out);
/* DBSCAN with min_samples > 1 (clustering.py:233-271: noise points exist then).  rhccq_eps_counts: counts[i] = number of palette points
 * within eps of point i, itself included (a core point has >= min_samples); rhccq_eps_border: labels_out[i] = core_label[i] for a core
 * point (its component's label from rhccq_eps_components on the core subset, in sklearn's order), else the smallest label among its
 * core neighbours, -1 (noise) if it has none.  thr / boundary / r2 as in rhccq_eps_components (rhccq_eps_threshold). */
int rhccq_eps_counts(rhccq_ctx* ctx, const uint32_t* keys, int32_t n, int32_t thr, int32_t boundary, double r2, int32_t* counts);
int rhccq_eps_border(rhccq_ctx* ctx, const uint32_t* keys, int32_t n, int32_t thr, int32_t boundary, double r2, const int32_t* core_label,
                     int32_t* labels_out);

/* ---- K2: per-cluster floor-mean colour (clustering.py:304-310,346-355) ----------------------
 * sums[k*4] (uint64 r,g,b,count) zero-initialised by the caller (one call per buffer: up to 2^24 points the sums are gathered
 * as two packed words per label and spread over the four fields at the end); labels in [0, k) or < 0 (skipped). */
int rhccq_cluster_sums(rhccq_ctx* ctx, const uint32_t* keys, const int32_t* labels, int64_t n, int64_t k,
                       unsigned long long* sums);
int rhccq_cluster_means(rhccq_ctx* ctx, const unsigned long long* sums, int64_t k, uint32_t* keys_out);

/* ---- K7: KMeans split of oversize clusters (clustering.py:720-775 -> sklearn KMeans) ---------
 * desc: int32[n_prob][6] = {off, n, k, rand_off, first_index, T}; rand: the MT19937 uniforms
 * (k-1)*T per problem at rand_off (numpy RandomState(42).uniform stream, generated on the host);
 * work: double workspace, 8*k doubles per problem at 8*koff (koff = desc-order prefix of k, given
 * as koff int64[n_prob]); labels_out int32 at the same offsets as keys; info int32[n_prob][4] =
 * {n_iter, strict, relocations, reserved}. */
int rhccq_kmeans(rhccq_ctx* ctx, const uint32_t* keys, const int32_t* desc, const int64_t* koff,
                 const double* rand, int32_t n_prob, int32_t max_n, double* work, int32_t* labels_out,
                 int32_t* info);
/* ---- K8: MiniBatchKMeans branch (clustering.py:207-230 -> sklearn MiniBatchKMeans(n_clusters, batch_size=1000,
 * random_state=42, n_init='auto').fit_predict) -------------------------------------------------------------------
 * The fit is sklearn's, operation for operation: the RandomState(42) stream is replayed from its raw MT19937 words
 * (`words`, resident on the device), k-means++ runs over the init sample in draw order, batches are
 * randint(0, n, 1000), centre updates add the batch members in batch order, reassigned centres take the rows
 * choice(1000, replace=False) names.  Where sklearn keeps np.argsort(counts)[:500] -- an unstable sort over tied counts --
 * the slots numpy's scalar quicksort fills are taken (RHCCQ_OPT_REASSIGN_ORDER, csrc/k8_npysort.h): the fit equals
 * scikit-learn's untouched fit_predict under numpy's scalar sort kernels.  oracle.minibatch_kmeans_labels states the same. */
typedef struct rhccq_mbk_problem {
  int64_t off;        /* first key of the problem in keys[] */
  int64_t n;          /* number of points */
  int64_t k;          /* n_clusters */
  int64_t koff;       /* offset (in clusters) into centres / weights / work arrays */
  int64_t init_off;   /* offset into init_idx / perm */
  int64_t init_n;     /* init sample size */
  int64_t rand_off;   /* offset into rand (k-1)*T uniforms */
  int32_t first;      /* first centre (position in the init sample, draw order) */
  int32_t T;          /* n_local_trials */
} rhccq_mbk_problem;
/* Internal pruning index of rhccq_mbk_init: perm (device, int32, same offsets as init_idx) receives, per problem, the
 * positions 0 .. init_n-1 of its init sample ordered by (Morton code of the sampled colour, position).  init_idx
 * (device; problem i's sample rows in RandomState draw order at [init_off, init_off + init_n), the problems back to
 * back) is only read.  Replaces nothing in the reference: 64 Morton-consecutive samples form a compact colour box,
 * which is what lets the k-means++ kernel skip blocks exactly.  tmp: rhccq_mbk_order_bytes(sum init_n) bytes. */
int64_t rhccq_mbk_order_bytes(int64_t total_samples);
int rhccq_mbk_order(rhccq_ctx* ctx, const uint32_t* keys, const rhccq_mbk_problem* probs_host,
                    int32_t n_prob, const int32_t* init_idx, int32_t* perm, void* tmp, int64_t tmp_bytes);
/* out[idx[i]] = min(out[idx[i]], val[i]), i < count: the first raster position of a palette entry that several entries of the level
 * below map to (merging.py:77-79 carried through clustering.py:373-377).  Host code; RHCCQ_E_ARG for an index outside [0, n_out). */
int rhccq_scatter_min_host(int64_t n_out, const int32_t* idx, const int64_t* val, int64_t count, int64_t* out);
/* The bookkeeping behind one clustered palette whose clusters need no split (clustering.py:296-377), host code: from the member
 * sums of the k clusters (rhccq_cluster_sums layout) the floor-mean colour of every non-empty cluster in label order behind
 * `nblack` black rows (new_keys: room for nblack + k) and lut[label] = new palette index (uint16-valued like the reference's
 * mapping_array; empty clusters map to the first row).  Returns the number of non-empty clusters, -1 when a cluster holds more
 * than mc colours (it takes the KMeans split instead), -2 for a bad argument. */
int64_t rhccq_cluster_plan_host(const unsigned long long* sums, int64_t k, int64_t mc, int32_t nblack, uint32_t* new_keys, int32_t* lut);
/* merge_region_components_simple in palette space (encoder/compression/merging.py:8-120), host code: the components are painted
 * in REVERSED order, every component's entries in the order of their first raster positions fp; a colour gets the index of its
 * first appearance in that sequence (index 0 = canvas black) and the smallest first position of its members.  Entries with key
 * 0 or fp >= fp_none do not appear (lut 0).  gkeys / gfp: room for 1 + sum(counts); lut[c]: counts[c] ints.  Pure host code. */
int rhccq_merge_palettes_host(int32_t n_comp, const uint32_t* const* keys, const int64_t* const* fp, const int32_t* counts,
                              int64_t fp_none, uint32_t* gkeys, int64_t* gfp, int32_t* const* lut, int64_t* n_out);
/* RandomState.randint(0, n, size) of numpy's legacy generator replayed ON THE HOST from raw MT19937 words in host memory
 * (masked rejection, one word per attempt): sklearn MiniBatchKMeans draws its validation and init samples this way before
 * k-means++ (clustering.py:207-218 -> _kmeans.py MiniBatchKMeans.fit).  out: int32[size] or NULL (stream position only).
 * Returns the words consumed, -1 when the table ends first, -2 for a bad argument.  Pure host code, no ctx. */
int64_t rhccq_mt_randint_host(const uint32_t* words, int64_t n_words, int64_t pos, int64_t n, int64_t size, int32_t* out);
/* out[i] = the i-th double numpy's legacy RandomState.uniform(size=count) / random_sample() yields when its
 * MT19937 stream stands at raw word `pos`: ((w[pos+2i] >> 5) * 2^26 + (w[pos+2i+1] >> 6)) / 2^53.  words: the raw
 * 32-bit outputs of MT19937(seed 42), resident on the device (roibasedimagecompression_amd/mt.py generates them
 * with numpy itself).  Serves the uniform(size=n_local_trials) draws of k-means++ (sklearn `_kmeans_plusplus`). */
int rhccq_mt_uniforms(rhccq_ctx* ctx, const uint32_t* words, int64_t pos, int64_t count, double* out);
/* greedy k-means++ (sklearn _kmeans_plusplus) on the init sample in exact integers, candidates searched over the
 * cumulative closest-distance sums in DRAW order; writes centres[(koff+j)*4 + {0,1,2}] (doubles, raw 0..255
 * coordinates; [3] = squared norm) and chosen[koff+j] (position in the init sample, draw order) */
int rhccq_mbk_init(rhccq_ctx* ctx, const uint32_t* keys, const rhccq_mbk_problem* probs_host,
                   int32_t n_prob, const int32_t* init_idx, const int32_t* perm, const double* rand,
                   double* centres, int32_t* chosen);
/* np.argsort(w)[:cap] AS A SET under numpy's scalar sort kernel (numpy/_core/src/npysort/quicksort.cpp aquicksort_<double>,
 * heapsort.cpp behind its depth limit) -- the selection inside a capped reassignment of rhccq_mbk_steps, exposed for tests.
 * w: double[k] on the device, non-negative integers < 2^32; 0 < cap < k; depth0 < 0 = numpy's depth limit 2 floor(log2 k)
 * (tests lower it to drive the heapsort branch); use_lds != 0: the part that still matters moves to LDS once it has shrunk to 7 680
 * elements (what rhccq_mbk_steps does), 0: global memory throughout; scratch: 16 k bytes; mask_out: uint32[(k + 31) / 32], bit j set
 * <=> j is among the first cap entries. */
int rhccq_npysort_head(rhccq_ctx* ctx, const double* w, int32_t k, int32_t cap, int32_t depth0, int32_t use_lds, void* scratch, uint32_t* mask_out);
/* run mini-batch steps step0 .. step0 + n_steps - 1 for every problem that has not stopped (call with step0 = 0 first -- that
 * call writes the problem tables at the head of `work`, later calls only queue kernels -- then with the number of steps launched so far; all problems of a call sequence share the step index).  state:
 * double[n_prob][16] = {[0] ewa, [1] ewa_min, [2] no_improvement, [3] samples since the last reassignment, [4] why the
 * problem stopped (0 running, 1 converged, 2 out of steps, 3 word table exhausted -- fatal: size `words` so that it
 * cannot happen; 4 sharded chain hand-off lost; 5 overlapped schedule diverged), [5] steps done, [6] have_ewa, [7] have_min, [8] zero-weight centres (initialise to k), [9] MT cursor
 * = raw words consumed so far (initialise to the position behind the k-means++ uniforms), [10] first batch drawn, [11]
 * stop_at (steps done when it stopped, 0 while running), [12..14] the odd-step twins of [3], [8], [9] (a step reads the
 * slots of its parity and writes the other's; after s steps the current values sit in the slots of parity s & 1)};
 * initialise everything but [8] and [9] to 0.  weights double[sum k]; words / n_words: raw MT19937(42) words on the
 * device -- a step consumes at most 16 384 of them. */
int rhccq_mbk_steps(rhccq_ctx* ctx, const uint32_t* keys, const rhccq_mbk_problem* probs_host,
                    int32_t n_prob, int64_t step0, int32_t n_steps, const uint32_t* words, int64_t n_words,
                    double* centres, double* weights, double* state, void* work, int64_t work_bytes,
                    int32_t estep_mode, int32_t estep_split);
/* estep_mode: how the batch E-step finds each point's nearest centre -- identical results either way:
 * RHCCQ_ESTEP_TILES brute force over LDS tiles of centres (few problems in flight), RHCCQ_ESTEP_GRID centres
 * re-binned into a 32^3 grid every step and searched ring by ring (many problems in flight: a batch of frames),
 * RHCCQ_ESTEP_AUTO grid when the problems handed over hold >= 200 000 centres.
 * estep_split (tiled E-step only; 0 or 1, 2, 4, 8): threads sharing one batch point, each walking 1/split of a
 * tile's centres -- shortens the per-thread chain when only a straggler problem is still running; same results. */
#define RHCCQ_ESTEP_AUTO 0
#define RHCCQ_ESTEP_TILES 1
#define RHCCQ_ESTEP_GRID 2
/* ORed into estep_mode: the caller has read the state and knows that no running problem reassigns during this call (state[8] /
 * [13] == 0 and fewer than 10 k samples since the last reassignment throughout): the second launch of a reassigning step
 * (the centres move, the next batch is drawn behind the shuffle) is left out; a problem that wants to reassign all the same
 * stops with state[4] = 5 */
#define RHCCQ_STEPS_NO_REASSIGN 0x100
int64_t rhccq_mbk_work_bytes(const rhccq_mbk_problem* probs_host, int32_t n_prob);
/* The same steps for ONE problem (n_prob == 1) that has no zero-weight centre left (state[8] / [13] == 0), with the batch
 * E-step of step t + 1 started BESIDE the update of step t: a step that does not reassign changes only the <= 1000 centres its
 * batch touched, so the next batch is first compared with every untouched centre (same launch as the update) and then with
 * the touched ones at their new values -- the same distances and the same first arg-min, off the chain of dependent
 * launches (a 4K frame ends in one problem that runs all its ~2000 steps: sklearn MiniBatchKMeans.fit,
 * clustering.py:207-218).  since0: "samples since the last reassignment" as step0 sees it (state[3] for an even step0,
 * state[12] for an odd one): the host derives from it which steps reassign (those run the classic E-step); a device state
 * that disagrees stops the problem with state[4] = 5.  *carry: in/out, 0 before the first call and after any rhccq_mbk_steps
 * call; bit 0 = the batch of step0 + 1 is drawn, bit 1 = the speculative tile minima of step0 are in `work`.  A step may
 * draw one batch ahead: size `words` for n_steps + 3 steps.  State, work and results as rhccq_mbk_steps (bit-identical). */
int rhccq_mbk_steps_overlapped(rhccq_ctx* ctx, const uint32_t* keys, const rhccq_mbk_problem* probs_host,
                               int32_t n_prob, int64_t step0, int32_t n_steps, const uint32_t* words, int64_t n_words,
                               double* centres, double* weights, double* state, void* work, int64_t work_bytes,
                               int32_t estep_split, int64_t since0, int32_t* carry);
/* final E-step over all points: labels_out int32 at the key offsets (first arg-min of
 * csq_j + (-2 * dot), brute force order-independent; uses a centre grid for pruning) */
int rhccq_mbk_assign(rhccq_ctx* ctx, const uint32_t* keys, const rhccq_mbk_problem* probs_host,
                     int32_t n_prob, const double* centres, void* work, int64_t work_bytes,
                     int32_t* labels_out);

/* ---- the fused frame encoder as ONE native entry ----------------------------------------------------------
 * rhccq.ipynb:978-1039 for one frame whose ROI / non-ROI segment label maps are given: per segment crop + black fix + unique colours
 * (subregions.py:315-449), cluster(q) per segment and merge per region (:634-679), per class merge on the frame canvas + cluster(2q)
 * (regions.py:9-70), merge of the classes + cluster(q3) + index dtype (image.py:243-286, compression.py:360-372) -- the ordering rules
 * of clustering.py:249-377 and merging.py:52-82 as native host code (csrc/encode_frame.hip), the kernels above underneath.  The region
 * classes run as host threads with HIP streams of their own, the MiniBatchKMeans problems of a class side by side on further streams;
 * nothing of the interpreter is on the path.  Result identical to roibasedimagecompression_amd.frame.FrameEncoder.encode (tests).
 *   classes[c]: labels = DEVICE int32[H*W] (0 = pixel not in the class, s >= 1 = segment id, ids global within the class and ascending
 *     inside each region); seg_region = HOST int32[n_seg] (region of segment id s at [s - 1]); region_bbox = HOST int32[n_region][4]
 *     (minr, minc, maxr, maxc); quality = the class's level-1 quality.  Classes in precedence order (ROI first).
 *   palette_out: HOST uint8[pal_cap][3]; indices_out: DEVICE buffer of H * W * 4 bytes, written as uint8 / uint16 / uint32 [H][W]
 *     (res->index_bytes) on the context's stream, complete when the call returns; n_unique_out (HOST int64[n_jobs], may be NULL):
 *     unique colours per segment.  At most 2048 segments per frame (beyond: the sort-based path of the Python FrameEncoder).
 * Returns 0, RHCCQ_E_ARG, RHCCQ_E_HIP or RHCCQ_E_LIMIT (too many segments, or pal_cap too small: res->n_colours says how many).
 * One call at a time per context (the context keeps the lanes' streams and device arenas between frames; use one context per host
 * thread that encodes); the call returns with every lane idle. */
typedef struct rhccq_class_desc {
  const int32_t* labels;
  int32_t n_seg;
  int32_t n_region;
  const int32_t* seg_region;
  const int32_t* region_bbox;
  int32_t quality;
  int32_t reserved;
} rhccq_class_desc;
typedef struct rhccq_frame_result {
  int32_t n_colours;       /* palette entries */
  int32_t index_bytes;     /* 1, 2 or 4 */
  int32_t shape[2];        /* (H, W), or the single component's box when only one component reaches level 3 (merging.py:16-21) */
  int32_t top_left[2];
  int32_t quality3;
  int32_t n_jobs;
  double ms[8];            /* host clocks: [0] scan [1] unique [2] levels 1-2 (slowest class) [3] level 3 [4] compose [5] remap [6] total */
  double class_ms[4][4];   /* per class (first four): level-1 clustering, first positions + merges, level-2 clustering, level-2 finish */
} rhccq_frame_result;
int rhccq_encode_frame(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, const rhccq_class_desc* classes, int32_t n_classes,
                       uint8_t* palette_out, int32_t pal_cap, void* indices_out, int64_t* n_unique_out, rhccq_frame_result* res);

/* ---- K6: index remap gather (clustering.py:373-377) ------------------------------------------ */
int rhccq_remap(rhccq_ctx* ctx, const int32_t* idx, int64_t n, const int32_t* lut, int64_t lut_n,
                int32_t* out);
/* fused final remap of a frame: class precedence = class order; value = lut[pal_off[job]+rank], then
 * lut2[value] when lut2 != NULL (lut = level-1 mapping, lut2 = the composed levels 2-3 over the clustered
 * palettes); values < 0 are transparent; out_elem_bytes in {1,2,4} */
int rhccq_frame_remap(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t n_class,
                      const int32_t* const* labels_host, const int32_t* job_base_host,
                      const uint32_t* bitmaps, const uint32_t* word_prefix, const int64_t* pal_off,
                      const uint32_t* fix_key, const int32_t* lut, const int32_t* lut2, int32_t default_index,
                      void* out, int32_t out_elem_bytes);

/* ---- K5: merge_region_components_simple (encoder/compression/merging.py:8-120) ---------------
 * one component at a time (the host assigns first-seen global indices between the two calls):
 * first_pos[idx] = min raster position (component raster) of in-canvas pixels showing palette
 * entry idx (entries >= pal_n are skipped, merging.py:72); paint writes lut[idx] where >= 0. */
int rhccq_merge_firstpos(rhccq_ctx* ctx, const int32_t* idx, int32_t h, int32_t w, int32_t top, int32_t left,
                         int32_t canvas_h, int32_t canvas_w, int32_t pal_n, int32_t* first_pos);
int rhccq_merge_paint(rhccq_ctx* ctx, const int32_t* idx, int32_t h, int32_t w, int32_t top, int32_t left,
                      int32_t canvas_h, int32_t canvas_w, const int32_t* lut, int32_t pal_n, int32_t* canvas);

/* ---- decode: palette[index] LUT gather (decoder/uncompression/uncompression.py:209) ---------- */
int rhccq_decode(rhccq_ctx* ctx, const void* idx, int32_t idx_elem_bytes, int64_t n, const uint8_t* palette,
                 int64_t pal_n, uint8_t* rgb_out);

/* ---- quality metrics (decoder/uncompression/comparison.py:30-80 calculate_quality_metrics) ------
 * a, b: uint8 RGB interleaved device images (4-byte aligned), n_pixels = H*W.
 * sums5 (device): [0..2] sum of squared differences per channel, [3] sum of |differences|, [4] max |difference|
 * -> MSE / RMSE / MAE / max error / per-channel MSE / PSNR (comparison.py:40,63-78) on the host. */
int rhccq_error_sums(rhccq_ctx* ctx, const uint8_t* a, const uint8_t* b, int64_t n_pixels, uint64_t* sums5);
/* calculate_adaptive_quality_metrics (comparison.py:345-536): tab768 (device, uint64[256][3]) row e = {pixels, sum of squared
 * differences over the 3 channels, sum of |differences|} of the pixels whose largest channel error is e; maxerr (device,
 * uint8[n_pixels], may be NULL) = that largest channel error per pixel (the outlier mask for the masked SSIM).  Percentiles,
 * outlier thresholds, the metrics of every pixel subset the reference forms and its error histogram follow on the host. */
int rhccq_error_tables(rhccq_ctx* ctx, const uint8_t* a, const uint8_t* b, int64_t n_pixels, uint64_t* tab768, uint8_t* maxerr);
/* structural_similarity(a, b, data_range=255, channel_axis=2, win_size=7) (comparison.py:47-49; algorithm of
 * scikit-image, unpinned by the reference): partial (device) receives, per workgroup tile, the sum over its
 * window centres of S for each channel; mean SSIM = sum(partial) / ((H-6)*(W-6)) averaged over the channels.
 * rhccq_ssim7_blocks gives the number of tiles (host only). */
int64_t rhccq_ssim7_blocks(int32_t H, int32_t W);
int rhccq_ssim7_sums(rhccq_ctx* ctx, const uint8_t* a, const uint8_t* b, int32_t H, int32_t W, double* partial,
                     int64_t n_blocks);

/* ---- split score of a region (encoder/subregions/split_score.py:15-142 calculate_split_score) ------------
 * One pass over the region image (uint8 RGB interleaved, device) and its mask (uint8, 0 / non-0, may be NULL: then
 * gray > 0.01 as the reference does): gray + CIE-Lab per pixel (scikit-image's rgb2gray / rgb2lab, float64), Sobel magnitude
 * of the four planes ('reflect' borders), uniform LBP(8,1) code of the gray plane (zeros outside the image).
 * partial (device, double[n_blocks][12]) receives per workgroup tile the masked sums {count, L, L^2, a, a^2, b, b^2,
 * sum over Lab planes of sqrt(2 sobel^2), sobel(gray), sobel(gray)^2, gray, gray^2}; hist42 (device, int32[42]) the LBP
 * histogram (10 bins) followed by the intensity histogram (32 bins over [0, 1]).  The host turns them into the three
 * scores (roibasedimagecompression_amd/api/split_score.py).  PARITY UNPINNED: scikit-image is absent from the build
 * container; the algorithms are restated from their published definitions. */
int64_t rhccq_split_stats_blocks(int32_t H, int32_t W);
int rhccq_split_stats(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, const uint8_t* mask, double* partial,
                      int64_t n_blocks, int32_t* hist42);

/* ---- masked SLIC (encoder/subregions/slic.py:41-104 -> skimage.segmentation.slic(mask=...); SURVEY 8f-2, parity unpinned)
 * rhccq_slic_assign: one assignment sweep of _slic_cython on a 2-D image: img (device, double[H][W][3], the Gaussian-smoothed
 * Lab image times 1 / compactness), mask (device u8), seg (device, double[K][5] = y, x, c0, c1, c2), step = the seed spacing;
 * labels (device int32[H][W]) = 1 + index of the nearest centroid whose window [c - 2 step, c + 2 step] holds the pixel
 * (first on ties), 0 outside the mask / in no window.  ignore_color != 0: spatial term only (the seed-relaxation sweeps).
 * rhccq_slic_connectivity_host: _enforce_label_connectivity_cython, a serial raster scan with breadth-first floods, as a
 * native HOST routine (both pointers are host memory; no context, no GPU). */
int rhccq_slic_assign(rhccq_ctx* ctx, const double* img, const uint8_t* mask, const double* seg, int32_t H, int32_t W, int32_t K,
                      double step, int32_t ignore_color, int32_t* labels);
int rhccq_slic_connectivity_host(const int32_t* labels_host, int32_t H, int32_t W, int32_t min_size, int32_t max_size,
                                 int32_t* out_host);

/* resize of enhanced_slic_with_texture (slic.py:42-44,82,101 -> skimage.transform.resize): scipy.ndimage.gaussian_filter1d along one axis of
 * a device float64 array viewed as [outer][len][inner] (mode 'mirror'; weights: device double[radius + 1] = centre, then offsets
 * 1..radius; scipy's summation order), scipy.ndimage.zoom order 1 / 0 with the per-axis index and weight tables computed on the host
 * (yi, wy: [2][oh]; xi, wx: [2][ow] for order 1; [oh], [ow] for order 0); zoom_linear clips to [lo, hi] as resize does. */
int rhccq_gauss1d_f64(rhccq_ctx* ctx, const double* in, int64_t outer, int32_t len, int64_t inner, const double* weights, int32_t radius, double* out);
int rhccq_zoom_linear_f64(rhccq_ctx* ctx, const double* in, int32_t H, int32_t W, int32_t C, const int32_t* yi, const double* wy, const int32_t* xi,
                          const double* wx, int32_t oh, int32_t ow, double lo, double hi, double* out);
int rhccq_zoom_nearest(rhccq_ctx* ctx, const void* in, int32_t elem_bytes, int32_t H, int32_t W, int32_t C, const int32_t* yi, const int32_t* xi,
                       int32_t oh, int32_t ow, void* out);

/* ---- ROI stage, region extraction (encoder/ROI/roi.py)
 * Connected components of a binary mask with statistics: what the reference takes from
 * cv2.connectedComponentsWithStats (roi.py:285-294 extract_connected_regions_fast, :229 fuse_adjacent_regions_optimized, and every
 * clean-up step of the encoder/ROI modules).  PARITY UNPINNED for the label NUMBERING: OpenCV is absent from the build container; its
 * published block-based algorithms (Grana's BBDT / Bolelli's Spaghetti, 8-connectivity) number components by their first
 * 2x2 block in block-raster order, the pixel-based 4-connectivity one by their first pixel; the partition itself is unambiguous.
 * rhccq_ccl: mask (device u8[H][W], nonzero = foreground), connectivity 4 or 8; numbering 0 = OpenCV's (above), 1 = by first
 *   pixel in raster order (scipy.ndimage.label / skimage.measure.label, roi.py:262 extract_connected_regions), 2 = unordered ids
 *   1..count and NO statistics (stats may be NULL; cap unused): enough to tell components apart, e.g. for Canny's hysteresis;
 *   work: rhccq_ccl_work_bytes(H, W, cap) bytes of device scratch; labels (device int32[H][W]): 0 = background, 1..count;
 *   stats (device int32[cap + 1][5]): row l = {CC_STAT_LEFT, TOP, WIDTH, HEIGHT, AREA} of label l, row 0 = the background;
 *   count (device int32): number of components.  When count > cap the labels are all 0 and the statistics meaningless:
 *   call again with cap >= count.  No host synchronisation inside.
 * rhccq_ccl_select: out[p] = lut[labels[p]] (u8 look-up per label: keep / drop whole components).
 * rhccq_roi_buffer = extract_roi_nonroi (roi.py:685-718): region_map (device u8, 1 = ROI core, 0 = non-ROI core), both cores
 * dilated buffer_size times with scipy's default cross (= L1 ball, border value 0), buffer zone = both dilations; outputs the
 * two masks (u8 0/1) and the two masked copies of rgb.  This one is pinned: the reference calls scipy.ndimage.binary_dilation. */
int64_t rhccq_ccl_work_bytes(int32_t H, int32_t W, int32_t cap);
int rhccq_ccl(rhccq_ctx* ctx, const uint8_t* mask, int32_t H, int32_t W, int32_t connectivity, int32_t numbering, void* work, int64_t work_bytes,
              int32_t cap, int32_t* labels, int32_t* stats, int32_t* count);
int rhccq_ccl_select(rhccq_ctx* ctx, const int32_t* labels, const uint8_t* lut, int64_t n_pixels, uint8_t* out);
/* tile-parallel labelling (one tile per GPU, seams stitched through one all-gather: roibasedimagecompression_amd/parallel.py tiled_ccl):
 * keys[l] (device u32[n_labels + 1], 0xffffffff where a label has no pixel) = the ordering key of component l of a TILE whose top-left
 * pixel sits at (y0, x0) of a frame frame_w pixels wide, in FRAME coordinates -- the key rhccq_ccl numbers by (numbering 0 with
 * connectivity 8: first 2x2 block in block-raster order, even y0 / x0 required; otherwise first pixel in raster order).  A component
 * that several tiles share takes the minimum of its parts' keys. */
int rhccq_ccl_keys(rhccq_ctx* ctx, const int32_t* labels, int32_t H, int32_t W, int32_t y0, int32_t x0, int32_t frame_w, int32_t numbering,
                   int32_t connectivity, int32_t n_labels, uint32_t* keys);
int rhccq_roi_buffer(rhccq_ctx* ctx, const uint8_t* region_map, const uint8_t* rgb, int32_t H, int32_t W, int32_t buffer_size,
                     uint8_t* roi_mask, uint8_t* nonroi_mask, uint8_t* roi_image, uint8_t* nonroi_image);

/* ---- ROI stage, edge front end (encoder/ROI/edges.py:35-71,173-195: get_edge_map = 20 adaptive threshold pairs scored on a
 * Canny edge map each + the final Canny; compute_local_density).  PARITY UNPINNED (OpenCV absent from the build container):
 * integer restatements of cvtColor(RGB2GRAY), Sobel 3x3, Canny (L1 gradient, aperture 3) -- csrc/edges.hip.
 * rhccq_edges_gray: gray (device u8[n]) + its 256-bin histogram (device int32[256]: Otsu's threshold on the host).
 * rhccq_edges_grad_hist: histogram (device int32[rhccq_edges_m2_bins()]) of gx^2 + gy^2, Sobel 3x3 with BORDER_REFLECT_101.
 * rhccq_canny_nms: img (device u8[H][W][channels], 1 or 3) -> nm (device u16[H][W]): Canny's L1 gradient magnitude where the
 *   pixel is a local maximum along its gradient direction, 0 elsewhere (threshold independent); mag_tmp u16[H*W], dxy_tmp int32[H*W].
 * rhccq_edges_above: mask[p] = nm[p] > low.
 * rhccq_label_reduce: red (device u64[n_labels + 1][4]) = per label {max of val16, sum of val8, sum of val8^2, pixel count} (either plane may
 *   be NULL: zeros).
 *   With rhccq_ccl on the mask this is Canny's hysteresis: a component is an edge iff its max exceeds `high`.
 * rhccq_box_count: out (device u16[H][W]) = number of non-zero pixels in the kernel_size x kernel_size window (odd, <= 31),
 *   BORDER_REFLECT_101: the integer content of compute_local_density's normalised box filter. */
int64_t rhccq_edges_m2_bins(void);
int rhccq_edges_gray(rhccq_ctx* ctx, const uint8_t* rgb, int64_t n_pixels, uint8_t* gray, int32_t* hist256);
int rhccq_edges_grad_hist(rhccq_ctx* ctx, const uint8_t* gray, int32_t H, int32_t W, int32_t* hist_m2);
int rhccq_canny_nms(rhccq_ctx* ctx, const uint8_t* img, int32_t H, int32_t W, int32_t channels, uint16_t* mag_tmp, int32_t* dxy_tmp,
                    uint16_t* nm);
int rhccq_edges_above(rhccq_ctx* ctx, const uint16_t* nm, int64_t n_pixels, int32_t low, uint8_t* mask);
int rhccq_label_reduce(rhccq_ctx* ctx, const int32_t* labels, const uint16_t* val16, const uint8_t* val8, int64_t n_pixels,
                       int32_t n_labels, uint64_t* red);
/* Canny's hysteresis verdict and the quality score's sums on the device: red = rhccq_label_reduce's (device u64[n + 1][4]);
 * out4 (device u64[4]) = {components whose max exceeds high, their pixels, their sum of val8, their sum of val8^2}; lut (device u8[n + 1], may
 * be NULL) = 255 for those components (for rhccq_ccl_select) */
int rhccq_edge_score(rhccq_ctx* ctx, const uint64_t* red, int32_t n_labels, int32_t high, uint64_t* out4, uint8_t* lut);
/* The scores of SEVERAL threshold pairs with nothing crossing to the host in between (find_best_edges_by_quality tries 20 pairs,
 * encoder/ROI/edges.py:40-71): per distinct `low` one mask + one labelling of {nm > low} (component count kept on the device) + one per-label
 * reduction capped at `cap` labels; per pair one verdict.  lows / highs: HOST int32[n_pairs], pairs with equal `low` adjacent.  out: DEVICE
 * uint64[n_pairs][5] = {edge components, edge pixels, sum gray, sum gray^2, components of {nm > low}}; a pair whose last number exceeds `cap`
 * must be scored through rhccq_label_reduce / rhccq_edge_score instead.  work: rhccq_canny_scores_bytes(H, W, cap) bytes. */
/* rhccq_canny_scores_nested: the same scores from ONE union-find grown over the thresholds in descending `low` (the sets {nm > low} nest):
 * every pixel is linked once over the whole search, a pair's verdict marks the roots of the pixels above `high` with a generation number and
 * sums the pixels under marked roots -- no component numbering, no per-label tables, no capacity.  Pairs in any order; out[i][4] = 0.
 * work: rhccq_canny_scores_nested_bytes(H, W) bytes. */
int64_t rhccq_canny_scores_nested_bytes(int32_t H, int32_t W);
int rhccq_canny_scores_nested(rhccq_ctx* ctx, const uint16_t* nm, const uint8_t* gray, int32_t H, int32_t W, const int32_t* lows_host,
                              const int32_t* highs_host, int32_t n_pairs, void* work, int64_t work_bytes, uint64_t* out);
int64_t rhccq_canny_scores_bytes(int32_t H, int32_t W, int32_t cap);
int rhccq_canny_scores(rhccq_ctx* ctx, const uint16_t* nm, const uint8_t* gray, int32_t H, int32_t W, const int32_t* lows_host,
                       const int32_t* highs_host, int32_t n_pairs, int32_t cap, void* work, int64_t work_bytes, uint64_t* out);
int rhccq_box_count(rhccq_ctx* ctx, const uint8_t* mask, int32_t H, int32_t W, int32_t kernel_size, uint16_t* out);
/* the same window, summing the pixel VALUES (maps that are not 0 / one value: the notebook's 0 / 1 / 255 planes) */
int rhccq_box_sum(rhccq_ctx* ctx, const uint8_t* plane, int32_t H, int32_t W, int32_t kernel_size, uint32_t* out);

/* ---- ROI stage, clean-up chain (encoder/ROI/{roi,small_regions,small_gaps,thin_regions2}.py): binary-mask operators.  PARITY
 * UNPINNED (OpenCV absent from the build container); masks are device u8 planes, set = non-zero, outputs 0 / 255.
 * rhccq_morph_dilate: dilation by a structuring element given as one half-width per row dy = -radius .. radius (-1 = empty row;
 *   radius <= 15), nothing set outside the image (cv2.dilate's default border); invert_in / invert_out = the erosion by the same
 *   symmetric element with cv2.erode's border rule (outside = set).  cv2.morphologyEx(MORPH_CLOSE) = dilate, then erode.
 * rhccq_mask_op: out = a | b (op 0), a & b (1), a & ~b (2), ~a (3).
 * rhccq_gap_bridge = bridge_small_gaps_fast (small_gaps.py:221-271): an unset pixel whose window count (rhccq_box_count; or window
 *   sum, rhccq_box_sum) is at least min_count is set when both opposite rays of one of the four direction pairs meet a set pixel within `reach` steps.
 * rhccq_dist_chamfer = cv2.distanceTransform(mask, DIST_L2, 3) in OpenCV's fixed point (16 fractional bits; hz_tmp: u16[H*W]).
 * rhccq_binary_sobel: m2 (device u8[H*W]) = gx^2 + gy^2 of the 3x3 Sobel of the 0/1 image (BORDER_REFLECT_101), max_out = its
 *   maximum (device int32); rhccq_lut_u8: out[p] = lut256[in[p]].
 * rhccq_label_sum: sums (device u64[n_labels + 1]) = per label (background 0 included) the sum of a u16 (value_bytes 2) or a
 *   non-negative int32 (value_bytes 4) plane. */
int rhccq_morph_dilate(rhccq_ctx* ctx, const uint8_t* in, int32_t H, int32_t W, int32_t radius, const int32_t* half_widths /* host */,
                       int32_t invert_in, int32_t invert_out, uint8_t* out);
/* the general form: rows dy = -up .. down, row i spans dx = -left[i] .. right[i] (both -1: empty row).  An even k x k OpenCV element
 * (protect_border_regions' default is 18, roi.py:824) has its anchor at k / 2: up = left = k / 2, down = right = k - 1 - k / 2. */
int rhccq_morph_dilate_spans(rhccq_ctx* ctx, const uint8_t* in, int32_t H, int32_t W, int32_t up, int32_t down, const int32_t* left /* host */,
                             const int32_t* right /* host */, int32_t invert_in, int32_t invert_out, uint8_t* out);
/* compute_local_density (edges.py:173-195) on ANY u8 plane for odd kernels up to 11 x 11 = OpenCV's direct filter2D path: float32 accumulator
 * over the taps in row-major order, BORDER_REFLECT_101; scale255 != 0: plane / 255.0 first.  (Binary planes go through rhccq_box_count and a table.) */
int rhccq_box_filter_seq(rhccq_ctx* ctx, const uint8_t* plane, int32_t H, int32_t W, int32_t kernel_size, int32_t scale255, float* out);
int rhccq_mask_op(rhccq_ctx* ctx, const uint8_t* a, const uint8_t* b, int64_t n, int32_t op, uint8_t* out);
int rhccq_gap_bridge(rhccq_ctx* ctx, const uint8_t* in, const void* counts /* u16 (count_bytes 2) or u32 (4) */, int32_t count_bytes, int32_t H,
                     int32_t W, int64_t min_count, int32_t reach, uint8_t* out);
int rhccq_dist_chamfer(rhccq_ctx* ctx, const uint8_t* mask, int32_t H, int32_t W, uint16_t* hz_tmp, int32_t* dist);
int rhccq_binary_sobel(rhccq_ctx* ctx, const uint8_t* mask, int32_t H, int32_t W, uint8_t* m2, int32_t* max_out);
int rhccq_lut_u8(rhccq_ctx* ctx, const uint8_t* in, const uint8_t* lut256, int64_t n, uint8_t* out);
/* out[p] = table[values[p]]: float32 table (device, n_table entries) indexed by a u16 plane (window counts -> densities) */
int rhccq_lut_u16_f32(rhccq_ctx* ctx, const uint16_t* values, const float* table, int32_t n_table, int64_t n, float* out);
int rhccq_label_sum(rhccq_ctx* ctx, const int32_t* labels, const void* values, int32_t value_bytes, int64_t n_pixels, int32_t n_labels,
                    uint64_t* sums);
/* hist (device u64[n_bins], n_bins <= 4096): histogram of the u16 plane over the pixels where mask is set (larger values ignored);
 * rhccq_value_mask: out = 255 where values[p] >= min_value and (mask is NULL or set), else 0 */
int rhccq_masked_hist(rhccq_ctx* ctx, const uint8_t* mask, const uint16_t* values, int64_t n_pixels, int32_t n_bins, uint64_t* hist);
int rhccq_value_mask(rhccq_ctx* ctx, const uint8_t* mask, const uint16_t* values, int64_t n_pixels, int32_t min_value, uint8_t* out);

/* ---- EXTENSION (no reference counterpart; named by BASELINE.json's north_star only): pixel-space DBSCAN ----
 * Features (x, y, L, a, b); q is a neighbour of p when dx^2 + dy^2 <= radius^2 (radius 0..4) and
 * dL^2 + da^2 + db^2 + spatial_weight^2 (dx^2 + dy^2) <= eps^2 (float32, operation order fixed in
 * csrc/px_dbscan_ext.hip and oracle.px_dbscan).  lin_lut: device float[256 + 2048]: 8-bit sRGB -> linear, then
 * 1024 pairs (f(i / 1024), f((i + 1) / 1024) - f(i / 1024)) of the Lab transfer function, interpolated linearly;
 * see Rhccq.px_tables().
 * rhccq_px_neighbours (the "fixed-radius neighbour pass": 3 B read + 4 B written per pixel): parent_out[p] = p for
 * a core pixel (>= min_pts neighbours, itself included), -1 otherwise; count_out (may be NULL) = neighbour count.
 * rhccq_px_expand (the "region growing"): lock-free union-find over the core pixels, then
 * labels_out[p] = 1 + smallest pixel index of p's cluster; non-core pixels take the smallest cluster among their
 * core neighbours, 0 = noise.  `parent` is the array rhccq_px_neighbours produced (modified in place). */
int rhccq_px_neighbours(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t radius, float eps,
                        float spatial_weight, int32_t min_pts, const float* lin_lut, int32_t* parent_out,
                        uint8_t* count_out);
int rhccq_px_expand(rhccq_ctx* ctx, const uint8_t* rgb, int32_t H, int32_t W, int32_t radius, float eps,
                    float spatial_weight, const float* lin_lut, int32_t* parent, int32_t* labels_out);

/* ---- EXTENSION (no reference counterpart): block DCT-II + region quantisation --
 * plane: float32[H*W]; block in {8,16}; qstep: float32 per tile [(H/block)*(W/block)];
 * coef_out float32[H*W] (may be NULL), q_out int16[H*W]. */
int rhccq_dct_quant(rhccq_ctx* ctx, const float* plane, int32_t H, int32_t W, int32_t block,
                    const float* qstep, float* coef_out, int16_t* q_out);
/* RGB u8 -> luma float32 (BT.601) plus per-tile qstep from a ROI mask (u8, may be NULL) */
int rhccq_luma_qstep(rhccq_ctx* ctx, const uint8_t* rgb, const uint8_t* roi_mask, int32_t H, int32_t W,
                     int32_t block, float q_roi, float q_bg, float* luma_out, float* qstep_out);

#ifdef __cplusplus
}
#endif
#endif /* RHCCQ_H */
