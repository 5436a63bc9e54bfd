/*
 * anorag.h — C ABI of libanorag_hip.so, the MI355X (gfx950) dense-retrieval hot path of AnoRAG.
 *
 * The reference (Kevinwu901113/ano-rag) is pure Python and has no FFI of its own; its hot path leaves
 * the repo at three third-party calls.  Every entry point below replaces one of those call sites (cited
 * per function as <reference file>:<line>).  The Python classes under ano-rag_amd/ (same names as the
 * reference's: EmbeddingManager / VectorIndex / VectorRetriever / HybridSearcher) are the only callers.
 *
 * Conventions
 *   - plain C types only; row-major float32 in/out, int64 ids, every buffer caller-allocated;
 *   - every function returns 0 on success and a negative ANR_E* code on failure; the message for the
 *     calling thread's last failure is anr_last_error();
 *   - "host" pointers are ordinary process memory, "_dev" pointers are HIP device memory of the handle's
 *     device; `stream` is a hipStream_t passed as void* (NULL = the handle's own stream);
 *   - a handle serialises its own calls internally (searches on one handle may be issued from several
 *     threads: reference query/query_processor.py:2761-2766 does exactly that).
 */
#ifndef ANORAG_H
#define ANORAG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANR_OK 0
#define ANR_EINVAL (-1)   /* bad argument                                   */
#define ANR_EHIP (-2)     /* a HIP runtime call failed (no device, OOM, …)  */
#define ANR_ESTATE (-3)   /* call not valid in the handle's current state   */
#define ANR_EINTERNAL (-4)

#define ANR_METRIC_IP 0 /* larger is better; cosine == IP on rows normalised at add time */
#define ANR_METRIC_L2 1 /* squared L2, smaller is better (faiss IndexFlatL2 convention)   */

const char *anr_last_error(void);
const char *anr_version(void);
/* number of visible HIP devices (0 when there is none); never fails */
int anr_device_count(void);

/* Device buffers for callers that hold no GPU runtime of their own (the Python classes pass the array sources of
 * anr_fuse_dense / the output of anr_bm25_scores_dev this way).  kind: 0 host->device, 1 device->host, 2 device->device. */
int anr_device_malloc(int32_t device, int64_t bytes, void **out);
int anr_device_free(int32_t device, void *ptr);
int anr_device_copy(int32_t device, void *dst, const void *src, int64_t bytes, int32_t kind);

/* ------------------------------------------------------------------------------------------------
 * Exact flat index: replaces faiss.IndexFlatIP / IndexFlatL2 as used by
 * vector_store/vector_index.py:77-80 (create), :187-196 (add), :223 (search), :415-426 (reset).
 * ---------------------------------------------------------------------------------------------- */
typedef struct anr_index anr_index;

/* dim: embedding dimension; metric: ANR_METRIC_*; normalize != 0 → rows and queries are L2-normalised
 * exactly as vector_index.py:265-282 does (zero-norm rows stay zero); device: HIP ordinal. */
int anr_index_create(int32_t dim, int32_t metric, int32_t normalize, int32_t device, anr_index **out);
int anr_index_destroy(anr_index *h);
/* pre-size device storage for n rows (optional; add grows geometrically otherwise) */
int anr_index_reserve(anr_index *h, int64_t n);
/* append n rows; sequential ids continue from ntotal (IndexFlat.add, vector_index.py:196) */
int anr_index_add(anr_index *h, const float *x_host, int64_t n);
int anr_index_add_dev(anr_index *h, const float *x_dev, int64_t n, void *stream);
int64_t anr_index_ntotal(const anr_index *h);
int32_t anr_index_dim(const anr_index *h);
int anr_index_reset(anr_index *h);
/* copy the stored (already preprocessed) float32 rows [i0, i0+n) back to the host */
int anr_index_reconstruct(anr_index *h, int64_t i0, int64_t n, float *out_host);

/* Diagnostics of the image the streaming scan reads (what pins ANR_OPT_SCAN_BITS 12 in the tests): rows [i0, i0 + n) of the
 * f16 image (bits = 16) or of the 12-bit image (bits = 12; ANR_ESTATE when the index keeps none) decoded to float32, and the
 * statistics the certificate uses — the largest row norm and the largest ||image row - stored float32 row|| of either image
 * (max_err12 = 0 without a 12-bit image) — with the image batches currently scan (12 or 16).  Any pointer may be NULL. */
int anr_index_reconstruct_scan_image(anr_index *h, int32_t bits, int64_t i0, int64_t n, float *out_host);
int anr_index_scan_image_stats(anr_index *h, int32_t *bits_in_use, float *max_norm, float *max_err16, float *max_err12);

/* Search nq queries for the k best rows (vector_index.py:223).  D[nq*k] scores best-first (inner
 * product, or squared L2), I[nq*k] row ids, padded with -1 (and -FLT_MAX / +FLT_MAX scores) when
 * k > ntotal, the faiss convention vector_index.py:234 relies on.  Ties in score are ordered by
 * ascending id.  Results are the exact top-k of the stored float32 rows.  k <= 1024 runs the streaming
 * pipeline; larger k (up to 2^20, synchronous calls only) computes every row's exact score and sorts on the device. */
int anr_index_search(anr_index *h, const float *q_host, int64_t nq, int32_t k, float *D, int64_t *I);
/* queries already in device memory (the output of anr_encoder_forward_dev), results to host buffers */
int anr_index_search_devq(anr_index *h, const float *q_dev, int64_t nq, int32_t k, float *D, int64_t *I);
/* same with device buffers, asynchronous on `stream` apart from one small status read-back */
int anr_index_search_dev(anr_index *h, const float *q_dev, int64_t nq, int32_t k, float *D_dev,
                         int64_t *I_dev, void *stream);

/* Asynchronous form: enqueues the search and returns.  Up to three batches are in flight (rotating workspaces), so
 * successive calls queue back to back on the device with no host synchronisation in between; `stream` is the stream the
 * queries were produced on (the batch waits for it when it is busy; NULL is the handle's own stream, NOT the legacy null
 * stream — queries produced on the null stream are handed over through a stream that waits for it, as
 * anorag_hip.sharded.ShardedStream.submit does).  D_dev / I_dev of calls still in flight must be
 * distinct buffers.  Results may be READ only after anr_index_wait() / anr_index_sync() have retired the batch: retiring
 * waits for it, runs the exact path for the queries whose certificate failed (patching D_dev / I_dev) and folds the
 * statistics.  (Until round 4 every call also made `stream` wait for the batch — ANR_OPT_STREAM_WAIT 1 restores that; the
 * marker ordered nothing that the retire does not, and in the device's four shared hardware queues it sat in front of the
 * NEXT batches' kernels: 0.348 -> 0.314 ms per batch at the 8-GPU shard size without it.) */
int anr_index_search_dev_async(anr_index *h, const float *q_dev, int64_t nq, int32_t k, float *D_dev,
                               int64_t *I_dev, void *stream);
int anr_index_sync(anr_index *h);
/* Partial form of anr_index_sync(): retires the OLDEST in-flight batches (a batch = up to 64 queries of one
 * asynchronous call) until at most `keep` (0..2) are still in flight.  The retired batches' D_dev / I_dev rows are
 * FINAL on return, while the newer ones keep the device busy — what a caller that must hand results on (the
 * row-sharded exchange of bench.py / anorag_hip/sharded.py) uses to stay pipelined without ever merging a
 * batch whose certificate recovery has not run. */
int anr_index_wait(anr_index *h, int32_t keep);

/* Exact scores of given rows: out[q][j] = score(query q, stored row ids[q][j]) — the
 * gather(note_embeddings, ids) . q that query/query_processor.py:3543-3589 recomputes per candidate
 * (there by re-encoding every candidate).  Queries are preprocessed like search queries; ids outside
 * [0, ntotal) give NaN.  Opt-in helper: the reference never takes this path. */
int anr_index_score_rows(anr_index *h, const float *q_host, int64_t nq, const int64_t *ids_host,
                         int32_t per_query, float *out_host);

/* All-pairs similarity of two sets of embeddings: EmbeddingManager.compute_similarity,
 * vector_store/embedding_manager.py:586-629, as one tiled kernel.  a [m][d], b [n][d] float32 host buffers;
 * metric 0 = cosine (rows / (||row|| + 1e-8) on both sides in float32, then the dot product, :602-609),
 * 1 = dot (:620), 2 = euclidean as 1 / (1 + ||a - b||) (:613-616).  Products are accumulated in float64;
 * out [m][n] float64 (the caller casts to the reference's result dtype). */
int anr_similarity_matrix(int32_t device, const float *a_host, int64_t m, const float *b_host, int64_t n, int32_t d,
                          int32_t metric, double *out_host);

/* Self join: every pair i < j of stored rows whose inner product reaches `threshold` — for a cosine
 * index (normalize = 1) the thresholded upper triangle of the N x N similarity matrix that
 * graph/relation_extractor.py:769-782 forms in full and :604-608 scans pair by pair, without the matrix.
 * Inner-product metric only.  Writes up to `cap` pairs (i, j, exact f32 score of the stored rows; order
 * unspecified) and sets *n_pairs to the number of qualifying pairs.  If the device-side lists were too
 * small, nothing is written, *n_pairs is set to minus an upper bound on the number of pairs, and the call
 * returns ANR_OK: call again with cap >= that bound. */
int anr_index_self_join(anr_index *h, float threshold, int64_t cap, int64_t *I_host, int64_t *J_host,
                        float *S_host, int64_t *n_pairs);

/* tuning / introspection */
#define ANR_OPT_FORCE_EXACT 1     /* 1: skip the f16 scan, run the dense exact path for every query  */
#define ANR_OPT_OVERFETCH 2       /* candidates kept per query before the exact re-score (0 = auto)  */
#define ANR_OPT_SAMPLE_ROWS 3     /* rows of the threshold sample (0 = auto)                          */
#define ANR_OPT_CAND_CAP 4        /* per-query candidate buffer entries                               */
#define ANR_OPT_TIMING 5          /* 1: record HIP-event time of the scan kernel in stats             */
#define ANR_OPT_ADD_RAW 6         /* 1: adds store rows as given (already normalised: reloading a saved index) */
#define ANR_OPT_STREAMS 7         /* streams the in-flight batches rotate over, 1..3 (default 3; 1 = one batch strictly after the other) */
#define ANR_OPT_ID_OFFSET 8       /* added to every returned id: the first global row of this shard (default 0) */
#define ANR_OPT_TINY 9            /* 1 (default): host-buffer searches of <= 4 queries run as ONE kernel launch (exact f32 rows / f64
                                     accumulate; completion by a word in pinned memory) where that is measured faster than the
                                     pipeline: k <= 64 and a short slice of rows per workgroup (e.g. k = 10 up to ~120 k x 768);
                                     2: wherever the path is structurally able (k <= 128, <= 1024 rows per workgroup; for tests);
                                     0: never.  Identical results either way */
#define ANR_OPT_FUSED_POST 10     /* 1 (default): candidate select + exact re-score + finalize run as ONE kernel per batch of more than
                                     4 queries (one workgroup per query); 2: for every batch; 0: always the three separate
                                     launches (identical results) */
int anr_index_set_option(anr_index *h, int32_t opt, int64_t value);

#define ANR_OPT_SHADOW 11         /* 0 (default): off; 1: a batch enqueued while another is still in flight (a pipelined caller) uses the
                                     "shadow" forms of its side kernels — threshold sample, ladder select, candidate select,
                                     re-score, finalize as workgroups of 4 waves, <= 56 registers, <= 58 KiB LDS, which are placed
                                     BESIDE the resident workgroups of the previous batch's scan instead of waiting for them
                                     (identical results); 2: every batch.  Measured slower than the wide kernels (a wave beside the
                                     scan waits ~8 us per memory round trip behind the scan's own requests): kept as an option */

#define ANR_OPT_SCHEDULE 12       /* 0 (default): one stream per in-flight batch (ANR_OPT_STREAMS of them); 1: role streams — all query
                                     preparations / threshold samples on one stream, all main scans on a second (strictly one after
                                     the other, in submission order), all select / re-score / finalize kernels on a third, two events
                                     per batch (measured 5 % slower at the 8-GPU shard size) */
#define ANR_OPT_STREAM_WAIT 13    /* 0 (default): results are read after anr_index_wait / anr_index_sync; 1: anr_index_search_dev_async
                                     also makes the caller's stream wait for the batch (the behaviour up to round 3) */
#define ANR_OPT_SCAN_BITS 14      /* the image of the corpus the streaming pass reads: 16 = the f16 image; 12 = a second image
                                     with every stored f16 rounded to its top 12 bits (6 mantissa bits), 25 % fewer bytes per
                                     row — the pass is HBM-bound, so it is that much shorter; 0 (default) = 12 for indexes of
                                     262 144 rows and more (built at the first such search), 16 below.  The certificate takes
                                     the coarser image's error norm (tracked per index like the f16 one) and K' grows from 192
                                     to 256 at k = 100: the results stay the exact top-k of the float32 rows.  A corpus whose
                                     certificates keep failing at the largest K' (near-duplicate neighbourhoods) goes back to
                                     the f16 image by itself.  Costs 1.5 bytes per stored value of device memory */

typedef struct anr_search_stats {
  int64_t n_queries;        /* queries of the last search call                                        */
  int64_t n_fallback;       /* of those, whose certificate failed (answered from the lists / by a second scan) */
  int64_t n_from_lists;     /* of those, recovered from the first scan's candidate lists (no second scan)  */
  int64_t n_dense_exact;    /* of those, answered by the dense exact path (forced, overflow, tiny index) */
  int64_t n_candidates;     /* candidates emitted by the scan, summed over queries                    */
  int64_t n_overflow;       /* queries whose candidate buffer overflowed                               */
  int64_t scan_bytes;       /* algorithmic bytes streamed by the dominant scan kernel (all launches)   */
  int32_t overfetch;        /* candidates re-scored per query                                          */
  int32_t sample_rows;
  float scan_ms;            /* HIP-event time of the main scan launches (ANR_OPT_TIMING)              */
  float total_ms;           /* HIP-event time of the whole call on the stream (ANR_OPT_TIMING)        */
} anr_search_stats;
/* statistics of the last synchronous search, or accumulated over the asynchronous searches retired since
 * anr_index_reset_stats() */
int anr_index_last_stats(anr_index *h, anr_search_stats *out);
int anr_index_reset_stats(anr_index *h);

/* Time line of the most recent batches of the scan pipeline (diagnostics; the reference has no counterpart — it is what
 * bench.py prints to explain a slow batch).  Retires every batch in flight, then copies up to max_batches records, oldest
 * first, ANR_BATCH_LOG_FIELDS int64 words each:
 *   [0] batch sequence number   [1] host clock (ns, monotonic) when its enqueue began   [2] when the enqueue returned
 *   [3] host clock when it was retired
 *   device clock (ns) at: [4] end of the query preparation  [5] end of the threshold sample  [6] end of the ladder select
 *   [7] first / [8] last workgroup of the main scan started  [9] end of the main scan  [10] end of the candidate select
 *   [11] end of the batch's last kernel
 *   [12] bit 0: shadow-sized side kernels; bits 8..: queries in the batch
 *   [13] host time (ns) the batch's recovery passes took at its retire (0: every certificate held)
 * (fused post kernel: [10] is 0).  *dev_minus_host_ns (may be NULL) receives an estimate of device clock - host clock,
 * good to a few microseconds, so that the two sets of times can be laid on one axis. */
#define ANR_BATCH_LOG_FIELDS 14
int anr_index_batch_log(anr_index *h, int64_t *out, int32_t max_batches, int32_t *n_out, int64_t *dev_minus_host_ns);

/* In-place row normalisation on host memory through the device (vector_index.py:276-280): rows with
 * zero norm are left unchanged. */
int anr_normalize_rows(float *x_host, int64_t n, int32_t d, int32_t device);

/* Merge P partial top-k lists per query (row-sharded corpus): Dp/Ip are [P][nq][k]
 * device buffers holding GLOBAL ids, -1 padded; writes the best k per query to D/I [nq][k].
 * larger_is_better selects the order.  Ties → lower id first. */
int anr_merge_topk_dev(int32_t device, const float *Dp_dev, const int64_t *Ip_dev, int32_t P, int64_t nq,
                       int32_t k, int32_t larger_is_better, float *D_dev, int64_t *I_dev, void *stream);
/* Same, with the parts d_stride floats / i_stride int64 apart instead of back to back — e.g. the receive
 * buffer of ONE all-gather whose per-rank chunk is [nq*k f32 | nq*k i64] (bench.py). */
int anr_merge_topk_strided_dev(int32_t device, const float *Dp_dev, const int64_t *Ip_dev, int64_t d_stride,
                               int64_t i_stride, int32_t P, int64_t nq, int32_t k, int32_t larger_is_better,
                               float *D_dev, int64_t *I_dev, void *stream);
/* The same merge for partial lists that already sit in host memory ([P][nq][k], e.g. written there by P
 * single-process shard searches): BASELINE.json north_star "per-shard local top-k merged on the host". */
int anr_merge_topk_host(const float *Dp, const int64_t *Ip, int32_t P, int64_t nq, int32_t k,
                        int32_t larger_is_better, float *D, int64_t *I);

/* ------------------------------------------------------------------------------------------------
 * Score fusion: the arithmetic of HybridSearcher.fuse, retrieval/hybrid_search.py:34-103, for a batch
 * of nq queries on the device (float64, bit-identical to the Python reference).
 * Each query has four (id, score) lists in the order dense, bm25, graph, path; ids are integers (the
 * caller maps note ids to integers), unique inside a list and listed in the caller's order (the order
 * matters for rrf ties, hybrid_search.py:66).  All lists are concatenated: `ids`/`scores` hold the
 * entries, `offs` is [nq][5] with the entry offset of each of the four lists of a query and its end.
 * method 0 = "linear" (max-normalised weighted sum, :83-100), 1 = "rrf" (:64-82).
 * Outputs per query, best first: out_ids [nq][pool] (-1 padded), out_final [nq][pool],
 * out_src [nq][pool][4] = the raw score of the id in each source or NaN when absent, out_count [nq].
 * A query may carry at most 4096 list entries in total.
 * ---------------------------------------------------------------------------------------------- */
int anr_fuse_lists(int32_t device, int32_t method, int64_t nq, const int64_t *ids, const double *scores,
                   const int64_t *offs, const double *weights /*[4]*/, double rrf_k, int32_t pool,
                   int64_t *out_ids, double *out_final, double *out_src, int32_t *out_count);

/* N-array form of the same fusion (BASELINE.json north_star: "fused elementwise+argk kernel"): a source may be the
 * full-corpus score vector that bm25_scores() returns (utils/bm25_search.py:286-340: one score per note, N of them,
 * zeros included) instead of a short list.  An ARRAY source is a device buffer [nq][array_len] and stands for the list
 * [(0, a[0]), (1, a[1]), ...]: every id < array_len is present (NaN marks an absent id), list order = id order.  The
 * other sources are short (id, score) lists in host memory (list_offs [nq + 1], at most 1024 entries per query in
 * all) or absent (all pointers NULL).  linear: any of the four sources may be arrays; rrf: exactly one of
 * dense / bm25 / graph (every id's exact rank among all N entries is counted in the streaming pass; no sort).
 * Outputs as anr_fuse_lists, bit-identical to it on the equivalent lists; pool <= 1024.  Ties in `final`: rrf —
 * the reference's ranks-dict insertion order; linear — lower id first (the reference iterates a set there). */
typedef struct anr_fuse_source {
  const void *array_dev;     /* device [nq][array_len], or NULL                  */
  int64_t array_len;
  int32_t array_dtype;       /* 0 = float64, 1 = float32                          */
  const int64_t *list_ids;   /* host, concatenated over the queries, or NULL      */
  const double *list_scores;
  const int64_t *list_offs;  /* host [nq + 1], or NULL                            */
  const double *array_max_dev; /* device [nq], or NULL: the maximum of each query's row of array_dev (non-NaN entries),
                                  when the producer knows it (anr_bm25_scores_dev, anr_bm25_combine_fields): linear
                                  then needs no max pass over the array.  Must be exact — it is the reference's
                                  max(scores) normaliser (hybrid_search.py:26-32) */
  /* SPARSE form of an array source (array_dev NULL, array_len = N > 0): row q is given by its explicit entries
   * (sparse_ids_dev[q][i], sparse_scores_dev[q][i]), i < sparse_count_dev[q] <= sparse_cap, in any order, ids distinct and
   * < N; every other id < N holds 0.0 (an explicit NaN marks an absent id) — what anr_bm25_sparse_dev leaves on the device
   * for a query whose postings touch a few thousand of the N notes.  Results are bit-identical to the same row handed
   * over as a dense array; the corpus-wide stream is replaced by work proportional to the entries.  At most ONE source
   * may be sparse (dense / bm25 / graph), the others are then short lists.  sparse_cap <= 8192: rows in any order (sorted
   * in LDS here); 8192 < sparse_cap <= 65536: every row of the call ASCENDING BY ID (checked on the device, ANR_EINVAL
   * otherwise) — the form anr_bm25_sparse_dev produces beyond 6144 documents. */
  const uint32_t *sparse_ids_dev;   /* device [nq][sparse_cap], or NULL */
  const double *sparse_scores_dev;  /* device [nq][sparse_cap]          */
  const int32_t *sparse_count_dev;  /* device [nq]                      */
  int64_t sparse_cap;
} anr_fuse_source;
typedef struct anr_fuse_dense_stats {
  int64_t n_queries;
  int64_t scan_bytes;        /* algorithmic bytes of the streaming pass: sum of array_len * sizeof(element) per query
                                (a sparse source: sparse_cap * 12 per query, the rows' capacity)                      */
  int64_t n_candidates;      /* ids the streaming pass kept, summed over queries                                    */
  float scan_ms;             /* HIP-event time of the streaming kernels (max pass + scan; a sparse source: sort + prep
                                + stage) over all sub-batches                                                          */
} anr_fuse_dense_stats;
int anr_fuse_dense(int32_t device, int32_t method, int64_t nq, const anr_fuse_source *sources /*[4]*/,
                   const double *weights /*[4]*/, double rrf_k, int32_t pool, int64_t *out_ids, double *out_final,
                   double *out_src, int32_t *out_count, anr_fuse_dense_stats *stats /* may be NULL */);

/* Weighted RRF over lists of ANY length (HybridSearcher.fuse, method rrf, when two or three lists are longer than
 * anr_fuse_lists holds — e.g. a dense score and a bm25 score for every note; retrieval/hybrid_search.py:60-72 ranks each
 * list by a stable descending sort).  Inputs as anr_fuse_lists (same offs layout, ids unique inside a list) with ids in
 * [0, n).  Each ranked list is sorted on the device in its own order (stable radix sort: equal scores keep their list
 * positions), the finals of all n ids are formed in the reference's order of operations and ordered by (final descending,
 * the reference's ranks-dict insertion order); ids held by no dense / bm25 / graph list are dropped.  Outputs as
 * anr_fuse_lists.  A correctness path (a handful of n-wide sorts per query): with ONE long list use anr_fuse_dense, which
 * counts the ranks it needs in a single streaming pass. */
int anr_fuse_rrf_long(int32_t device, int64_t nq, const int64_t *ids_host, const double *scores_host, const int64_t *offs_host,
                      int64_t n, const double *weights /*[4]*/, double rrf_k, int32_t pool, int64_t *out_ids,
                      double *out_final, double *out_src, int32_t *out_count);

/* Candidate-level fusion of QueryProcessor (opt-in, the reference's own call path re-encodes
 * candidates instead): the scoring loops of _hybrid_search (query/query_processor.py:3703-3760) and of
 * _enhanced_hybrid_search_v2 (:1104-1143) for nq queries at once, candidates of query q = entries [offs[q], offs[q+1]).
 *   mode 0 linear: v = a; s = b; s *= 0.1 if flags&1 (misses the must-have terms); v *= 1.2 per boost entity found
 *                  (n_ent); s *= 1.3 per boost predicate found (n_pred); score = wa*v + wb*s
 *   mode 1 rrf:    0-based ranks of stable descending sorts of a and of b; score = wa/(rrf_k+ra) + wb/(rrf_k+rb);
 *                  score *= 0.1 if flags&1
 *   mode 2 v2:     f = 1.0*a + 0.6*b; f *= mult[0] (section); f *= mult[1] (lexical); if f < noise and flags&1 (must-have
 *                  terms NOT satisfied): f = 0; f *= mult[2] (entity boost); f *= mult[3] (predicate boost)
 * score[] per candidate (float64, the reference's order of operations) and order[] = the candidate indices of each
 * query (relative to its range) in the stable descending order the reference sorts into.  All arrays host memory;
 * flags / n_ent / n_pred may be NULL (= 0), mult is [total][4] and required for mode 2. */
int anr_fuse_candidates(int32_t device, int32_t mode, int64_t nq, const int64_t *offs, const double *a, const double *b,
                        const int32_t *flags, const int32_t *n_ent, const int32_t *n_pred, const double *mult,
                        double wa, double wb, double rrf_k, double noise, double *score, int32_t *order);

/* ------------------------------------------------------------------------------------------------
 * Sentence encoder: replaces SentenceTransformer.encode as called at
 * vector_store/embedding_manager.py:392-399 (and :357) — transformer forward + pooling + optional L2
 * normalisation for BERT-family models (bert / roberta / xlm-roberta / mpnet).  Tokenisation is host work.
 * ---------------------------------------------------------------------------------------------- */
typedef struct anr_encoder anr_encoder;
typedef struct anr_encoder_config {
  int32_t n_layers, hidden, n_heads, intermediate;
  int32_t vocab_size, max_positions, type_vocab_size;
  int32_t pos_offset; /* 0 for bert; padding_idx + 1 (= 2) for roberta / xlm-roberta position ids          */
  int32_t pooling;    /* 0 = mean over the attention mask, 1 = CLS token (sentence-transformers Pooling)    */
  int32_t act;        /* 0 = gelu (erf)                                                                      */
  float ln_eps;
} anr_encoder_config;

int anr_encoder_create(const anr_encoder_config *cfg, int32_t device, anr_encoder **out);
int anr_encoder_destroy(anr_encoder *e);
/* Upload one float32 tensor by name (row-major, nn.Linear weights as [out][in]):
 *   emb.word emb.pos emb.type emb.ln.g emb.ln.b
 *   L<i>.{q,k,v,o,ffn1,ffn2}.{w,b}   L<i>.ln1.{g,b} (after attention)   L<i>.ln2.{g,b} (after the FFN)
 *   rel.bias (optional, MPNet): [n_heads][2 P - 1], P = max_positions - pos_offset — added to the scaled
 *   attention score of (query i, key j) at index j - i + P - 1 (all-mpnet-base-v2, the reference's second
 *   fallback model, embedding_manager.py:218-219) */
int anr_encoder_set_tensor(anr_encoder *e, const char *name, const float *data, int64_t n_elements);
int anr_encoder_finalize(anr_encoder *e); /* fails if a tensor is missing */
/* ids [B][L] (right-padded), lengths [B] = number of real tokens per row (the attention mask), type_ids
 * [B][L] or NULL; out_host [B][hidden] float32.  normalize != 0 applies F.normalize(p=2, dim=1). */
int anr_encoder_forward(anr_encoder *e, const int32_t *ids, const int32_t *lengths, const int32_t *type_ids,
                        int32_t B, int32_t L, int32_t normalize, float *out_host);

/* The same forward with the embeddings left in DEVICE memory (caller-allocated on the encoder's device, float32
 * rows of `hidden`): sequence b goes to row out_rows[b] (host array [B]; NULL = row b), so the length-sorted batches
 * of SentenceTransformer.encode land in input order.  Feeds anr_index_add_dev / anr_index_search_dev directly — the
 * index build (vector_store/retriever.py:140-157) and the query path (:206-216) without the device -> host -> device
 * round trip per batch.  out_dev is complete when the call returns. */
int anr_encoder_forward_dev(anr_encoder *e, const int32_t *ids, const int32_t *lengths, const int32_t *type_ids,
                            int32_t B, int32_t L, int32_t normalize, float *out_dev, const int32_t *out_rows);
/* anr_encoder_forward for callers that share the handle from several threads with a query or a few each (the reference's
 * query-time pattern: main_musique.py:487-494, query/query_processor.py:2761-2766): calls that arrive while a forward is
 * running wait in a combining queue and are merged into the next forward — same normalize flag, same use of type_ids, at most
 * 2048 padded tokens per forward, so an embedding never depends on who else was in flight: bit-identical to
 * anr_encoder_forward.  Larger requests run alone, as anr_encoder_forward.  anr_encoder_shared_stats: forwards run and
 * requests served through the queue so far (either pointer may be NULL). */
int anr_encoder_forward_shared(anr_encoder *e, const int32_t *ids, const int32_t *lengths, const int32_t *type_ids,
                               int32_t B, int32_t L, int32_t normalize, float *out_host);
int anr_encoder_shared_stats(anr_encoder *e, int64_t *forwards, int64_t *requests);

/* ------------------------------------------------------------------------------------------------
 * BM25 scoring: SimpleBM25.get_scores / bm25_scores of utils/bm25_search.py:43-63,
 * :286-340 over a CSR postings image.  Tokenisation, term ids and the float64 posting weights
 *   w(t,d) = idf_t * (tf*(k1+1) / (tf + k1*(1 - b + b*(dl_d/avgdl))))
 * are host work (anorag_hip/bm25_search.py); the device adds them per query in the reference's order, so the
 * float64 scores are bit-identical.  normalize != 0 divides by the maximum when it is > 0 (:329-333).
 * A term's posting list holds each document once, document ids strictly ascending (checked at create).
 * ---------------------------------------------------------------------------------------------- */
typedef struct anr_bm25 anr_bm25;
int anr_bm25_create(int32_t device, int64_t n_docs, int64_t n_terms, const int64_t *indptr /*[n_terms+1]*/,
                    const int32_t *doc_ids, const double *weights, anr_bm25 **out);
int anr_bm25_destroy(anr_bm25 *h);
/* query i = term ids q_terms[q_indptr[i] .. q_indptr[i+1]) in query-token order (repeats count twice, unknown
 * tokens are simply omitted).  out_host: dense [nq][n_docs] float64. */
int anr_bm25_scores(anr_bm25 *h, int64_t nq, const int64_t *q_indptr, const int32_t *q_terms, int32_t normalize,
                    double *out_host);
/* the same scores left in DEVICE memory ([nq][n_docs] float64, caller-allocated on the handle's device): the array
 * source anr_fuse_dense takes — the N-vector never crosses PCIe.  max_dev (device [nq], may be NULL): the maximum of each
 * output row, a by-product of the scoring (a pass over the query's postings, not over n_docs) — hand it to
 * anr_fuse_source.array_max_dev and linear fusion needs no max pass of its own */
int anr_bm25_scores_dev(anr_bm25 *h, int64_t nq, const int64_t *q_indptr, const int32_t *q_terms, int32_t normalize,
                        double *out_dev, double *max_dev);
/* FieldWeightedBM25 (utils/bm25_search.py:66-146, :190-234): per field one anr_bm25 handle scores the queries
 * (anr_bm25_scores_dev, normalize = 0); this adds the fields up — total = sum_f weights[f] * field_scores[f], in field
 * order, float64, bit-identical to get_scores — and, with normalize != 0, divides by the per-query maximum when it is
 * > 0 (field_weighted_bm25_scores).  field_scores_dev: host array of n_fields (<= 8) device pointers [nq][n_docs];
 * out_dev [nq][n_docs] on the same device (it may be one of the inputs); max_dev as in anr_bm25_scores_dev. */
int anr_bm25_combine_fields(int32_t device, int32_t n_fields, const double *const *field_scores_dev,
                            const double *weights, int64_t nq, int64_t n_docs, int32_t normalize, double *out_dev,
                            double *max_dev);
/* The scores of anr_bm25_scores_dev in SPARSE form, left on the device for anr_fuse_dense (anr_fuse_source.sparse_*): per
 * query the documents its postings touch — ids_dev / scores_dev [nq][cap] (cap <= 65536), count_dev [nq], max_dev
 * [nq] (may be NULL; the maximum over all n_docs scores, the untouched zeros included).  The N-vector is never
 * materialised.  cap <= 6144: a workgroup accumulates its query in an LDS hash table, rows unordered; 6144 < cap <= 65536
 * (round 4): the row is cut into slices of the document range, one workgroup per slice (presence bits -> rank = slot, sums
 * in LDS), rows ASCENDING BY ID — what anr_fuse_dense asks of rows beyond 8192 entries.  Either way a document's additions
 * happen one at a time in the reference's token order, so every score equals anr_bm25_scores' bit for bit.  A query that
 * touches more than cap documents gets count -1 (its row is unusable: score that query with anr_bm25_scores_dev); so does
 * (cap > 6144) one whose summed posting lengths exceed 4 cap or one slice of which holds more than 4096 documents.
 * out_count_host (may be NULL): the counts, copied back. */
int anr_bm25_sparse_dev(anr_bm25 *h, int64_t nq, const int64_t *q_indptr, const int32_t *q_terms, int32_t normalize,
                        int32_t cap, uint32_t *ids_dev, double *scores_dev, int32_t *count_dev, double *max_dev,
                        int32_t *out_count_host);
/* sparse form for the fusion: the documents with a non-zero score, unordered; out_count may exceed cap
 * (the lists are then truncated) */
int anr_bm25_nonzero(anr_bm25 *h, int64_t nq, const int64_t *q_indptr, const int32_t *q_terms, int32_t normalize,
                     int32_t cap, int32_t *out_docs, double *out_scores, int32_t *out_count);

#ifdef __cplusplus
}
#endif
#endif /* ANORAG_H */
