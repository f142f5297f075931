/*
 * dsv2_hip.h -- C ABI of libdsv2hip.so, the MI355X (gfx950) implementation of the
 * DSV2 v2.8 per-frame encode/decode hot path.
 *
 * The library is a drop-in for the reference codec library: it exports the same
 * symbols with the same struct layouts, ownership rules and return codes, so the
 * reference's own CLI (src/dsv_main.c) links against it unchanged.
 * Each declaration cites the reference interface it replaces.
 *
 *   Section 1  shared types                  (reference src/dsv.h:100-273, dsv_internal.h:40-47)
 *   Section 2  encoder API                   (reference src/dsv_encoder.h:68-199)
 *   Section 3  decoder API                   (reference src/dsv_decoder.h:30-61)
 *   Section 4  frame / buffer / misc helpers (reference src/dsv.h:224-324)
 *   Section 5  hot-path seam on HOST buffers (reference src/dsv_internal.h:112-147, dsv.h:232-237)
 *              same signatures as the reference's internal functions; each call
 *              uploads its operands to HBM, runs the HIP kernels, downloads the result.
 *   Section 6  hot-path seam on DEVICE-resident buffers (dsv2hip_*), used by the
 *              encoder/decoder internally and by bench.py (inputs already in HBM).
 *
 * There is no CPU fallback: every compute entry point aborts with a message on
 * stderr when no HIP device is usable.
 */
#ifndef DSV2_HIP_H
#define DSV2_HIP_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* Section 1: shared types                                                   */

/* packet types, dsv.h:38-45 */
#define DSV_PT_META 0x00
#define DSV_PT_PIC 0x04
#define DSV_PT_EOS 0x10
#define DSV_PACKET_HDR_SIZE 14
#define DSV_PACKET_TYPE_OFFSET 5
#define DSV_PACKET_PREV_OFFSET 6
#define DSV_PACKET_NEXT_OFFSET 10

/* chroma subsampling codes, dsv.h:78-96: bits 3..2 = horizontal shift, bits 1..0 = vertical shift */
#define DSV_SUBSAMP_444 0x0
#define DSV_SUBSAMP_422 0x4
#define DSV_SUBSAMP_420 0x5
#define DSV_SUBSAMP_411 0x8
#define DSV_SUBSAMP_410 0xA
#define DSV_SUBSAMP_UYVY 0x14
#define DSV_FORMAT_H_SHIFT(f) (((f) >> 2) & 0x3)
#define DSV_FORMAT_V_SHIFT(f) ((f) & 0x3)

#define DSV_MIN_BLOCK_SIZE 16
#define DSV_MAX_BLOCK_SIZE 32
#define DSV_MAX_QP_BITS 12
#define DSV_MAX_QP ((1 << DSV_MAX_QP_BITS) - 1)

typedef uint32_t DSV_FNUM;

typedef struct { /* dsv.h:100-119 */
    int width, height, subsamp;
    int fps_num, fps_den;
    int aspect_num, aspect_den;
    int inter_sharpen;
    int reserved;
} DSV_META;

typedef struct { /* dsv.h:121-127 */
    uint8_t *data; /* pixel (0,0); bordered planes have 32 valid pixels on every side */
    int len;
    int format;
    int stride;
    int w, h;
} DSV_PLANE;

typedef int32_t DSV_SBC; /* subband coefficient */
typedef struct {         /* dsv.h:131-135 */
    DSV_SBC *data;
    int width, height;
} DSV_COEFS;

typedef struct { /* dsv.h:137-149 */
    uint8_t *alloc;
    DSV_PLANE planes[3];
    int refcount;
    int format;
    int width, height;
    int border;
} DSV_FRAME;

/* motion vector / per-block mode record, 16 bytes, dsv.h:171-216 */
typedef struct {
    union {
        struct {
            int16_t x, y; /* quarter-pel units */
        } mv;
        int32_t all;
    } u;
    uint32_t flags; /* DSV_MV_BIT_* */
    uint16_t err;
    uint16_t dc; /* bit 8 (DSV_SRC_DC_PRED): low byte is a transmitted DC */
    uint8_t submask;
} DSV_MV;

#define DSV_MV_BIT_INTRA 0
#define DSV_MV_BIT_EPRM 1
#define DSV_MV_BIT_MAINTAIN 2
#define DSV_MV_BIT_SKIP 3
#define DSV_MV_BIT_RINGING 4
#define DSV_MV_BIT_NOXMITY 5
#define DSV_MV_BIT_NOXMITC 6
#define DSV_MV_BIT_SIMCMPLX 7
#define DSV_SRC_DC_PRED 0x100
#define DSV_MASK_ALL_INTRA 0xF

typedef struct { /* dsv.h:239-266 */
    DSV_META *vidmeta;
    int effort;
    int do_psy;
    int is_ref;
    int has_ref;
    int blk_w, blk_h;
    int nblocks_h, nblocks_v;
    int temporal_mc;
    int lossless;
    int reserved;
} DSV_PARAMS;

typedef struct { /* dsv.h:268-271 */
    uint8_t *data;
    unsigned len;
} DSV_BUF;

typedef struct { /* dsv_internal.h:40-47 */
    DSV_PARAMS *params;
    DSV_MV *mvs;
    uint8_t *blockdata; /* per-block DSV_IS_* flag bytes */
    uint8_t cur_plane;
    uint8_t isP;
    DSV_FNUM fnum;
} DSV_FMETA;

typedef struct { /* dsv_internal.h:49-52, MSB-first bit cursor */
    uint8_t *start;
    unsigned pos;
} DSV_BS;

/* per-block flag bits in DSV_FMETA.blockdata, dsv_internal.h:96-110 */
#define DSV_IS_STABLE 0x01
#define DSV_IS_MAINTAIN 0x02
#define DSV_IS_SKIP 0x04
#define DSV_IS_RINGING 0x08
#define DSV_IS_INTRA 0x10
#define DSV_IS_EPRM 0x20
#define DSV_IS_SIMCMPLX 0x40

/* ------------------------------------------------------------------------- */
/* Section 2: encoder (dsv_encoder.h)                                        */

#define DSV_ENCODER_VERSION 14
#define DSV_GOP_INTRA 0
#define DSV_GOP_INF 0x7fffffff
#define DSV_ENC_NUM_BUFS 0x03
#define DSV_ENC_FINISHED 0x04
#define DSV_MIN_EFFORT 0
#define DSV_MAX_EFFORT 10
#define DSV_RATE_CONTROL_CRF 0
#define DSV_RATE_CONTROL_ABR 1
#define DSV_RATE_CONTROL_CQP 2
#define DSV_MAX_PYRAMID_LEVELS 5
#define DSV_RC_QUAL_SCALE 4
#define DSV_MAX_QUALITY 100
#define DSV_RC_QUAL_MAX (DSV_MAX_QUALITY * DSV_RC_QUAL_SCALE)
#define DSV_USER_QUAL_TO_RC_QUAL(u) ((u) * DSV_RC_QUAL_SCALE)

#define DSV_PSY_ADAPTIVE_QUANT (1 << 0)
#define DSV_PSY_CONTENT_ANALYSIS (1 << 1)
#define DSV_PSY_I_VISUAL_MASKING (1 << 2)
#define DSV_PSY_P_VISUAL_MASKING (1 << 3)
#define DSV_PSY_ADAPTIVE_RINGING (1 << 4)
#define DSV_PSY_ALL 0xff

struct DSV_STATS { /* dsv_encoder.h:116-147 */
    unsigned inum, pnum, iqual, pqual, iminq, pminq, imaxq, pmaxq;
    unsigned isize, psize, imins, pmins, imaxs, pmaxs;
    unsigned mb, mbI, mbP, mbdc, mbsub;
    unsigned mbsubs[4];
    unsigned eprm, skip;
    unsigned fpx, hpx, qpx, fpy, hpy, qpy;
    unsigned ifnum, pfnum;
};

struct DSV_STAB_ACC {
    int32_t x, y;
};

/* Configuration is by writing the public fields between dsv_enc_init and dsv_enc_start,
 * exactly as with the reference (dsv_encoder.h:68-188).  The fields after `stats` are
 * internal state; `ref` holds this library's device-side encoder context. */
typedef struct {
    int quality;
    int effort;
    int gop;
    int do_scd;
    int do_temporal_aq;
    int do_psy;
    int do_dark_intra_boost;
    int do_intra_filter;
    int do_inter_filter;
    int skip_block_thresh;
    int block_size_override_x;
    int block_size_override_y;
    int variable_i_interval;
    int rc_mode;
    unsigned bitrate;
    int rc_pergop;
    int min_q_step;
    int max_q_step;
    int min_quality;
    int max_quality;
    int min_I_frame_quality;
    int prev_I_frame_quality;
    int intra_pct_thresh;
    int scene_change_pct;
    unsigned stable_refresh;
    int pyramid_levels;
    struct DSV_STATS stats;

    unsigned rc_qual;
    unsigned rf_total;
    unsigned rf_reset;
    int rf_avg;
    int total_P_frame_q;
    int avg_P_frame_q;
    int prev_complexity;
    int curr_complexity;
    int curr_avgmot;
    int curr_intra_pct;
    int curr_scblocks;
    int prev_chaos;
    int motion_chaos;
    int motion_static;
    int avg_err;
    int auto_filter;

    void (*frame_callback)(DSV_META *m, DSV_FRAME *orig, DSV_FRAME *recon);

    DSV_FNUM next_fnum;
    void *ref; /* reference: DSV_ENCDATA*; here: opaque device encoder context */
    DSV_META vidmeta;
    int prev_link;
    int force_metadata;
    struct DSV_STAB_ACC *stability;
    unsigned refresh_ctr;
    uint8_t *blockdata;
    uint8_t *intra_map;
    DSV_FNUM prev_gop;
    int prev_quant;
} DSV_ENCODER;

void dsv_enc_init(DSV_ENCODER *enc);                         /* dsv_encoder.h:190, dsv_encoder.c:1320 */
void dsv_enc_free(DSV_ENCODER *enc);                         /* dsv_encoder.h:191 */
void dsv_enc_set_metadata(DSV_ENCODER *enc, DSV_META *md);   /* dsv_encoder.h:192 */
void dsv_enc_force_metadata(DSV_ENCODER *enc);               /* dsv_encoder.h:193 */
void dsv_enc_start(DSV_ENCODER *enc);                        /* dsv_encoder.h:195 */
/* consumes `frame` (one reference), returns 1 or 2 packets in bufs[] (metadata first);
 * the caller frees each with dsv_buf_free: dsv_encoder.h:198, dsv_encoder.c:1431 */
int dsv_enc(DSV_ENCODER *enc, DSV_FRAME *frame, DSV_BUF *bufs);
void dsv_enc_end_of_stream(DSV_ENCODER *enc, DSV_BUF *bufs); /* dsv_encoder.h:199 */

/* ------------------------------------------------------------------------- */
/* Section 3: decoder (dsv_decoder.h)                                        */

#define DSV_DECODER_VERSION 2
#define DSV_DRAW_STABHQ 1
#define DSV_DRAW_MOVECS 2
#define DSV_DRAW_IBLOCK 4

typedef struct { /* dsv_decoder.h:39-45; zero-initialised by the caller */
    DSV_META vidmeta;
    void *ref; /* reference: DSV_IMAGE*; here: opaque device decoder context */
    int draw_info;
    int got_metadata;
} DSV_DECODER;

#define DSV_DEC_OK 0
#define DSV_DEC_ERROR 1
#define DSV_DEC_EOS 2
#define DSV_DEC_GOT_META 3
#define DSV_DEC_NEED_NEXT 4

/* frees `buf`; on DSV_DEC_OK with a picture packet *out holds one reference the caller
 * releases with dsv_frame_ref_dec: dsv_decoder.h:54, dsv_decoder.c:394 */
int dsv_dec(DSV_DECODER *d, DSV_BUF *buf, DSV_FRAME **out, DSV_FNUM *fn);
DSV_META *dsv_get_metadata(DSV_DECODER *d); /* dsv_alloc'd copy, dsv_decoder.h:58 */
void dsv_dec_free(DSV_DECODER *d);          /* dsv_decoder.h:61 */

/* ------------------------------------------------------------------------- */
/* Section 4: helpers (dsv.h:224-324)                                        */

void dsv_mk_coefs(DSV_COEFS *c, int format, int width, int height);
DSV_FRAME *dsv_mk_frame(int format, int width, int height, int border);
DSV_FRAME *dsv_load_planar_frame(int format, void *data, int width, int height);
DSV_FRAME *dsv_frame_ref_inc(DSV_FRAME *frame);
void dsv_frame_ref_dec(DSV_FRAME *frame);
DSV_FRAME *dsv_clone_frame(DSV_FRAME *f, int border);
void dsv_plane_xy(DSV_FRAME *f, DSV_PLANE *out, int c, int x, int y);
void dsv_mk_buf(DSV_BUF *buf, int size);
void dsv_buf_free(DSV_BUF *buf);
void *dsv_alloc(int size); /* zero-initialised */
void dsv_free(void *ptr);
void dsv_memory_report(void);
void dsv_set_log_level(int level);
int dsv_get_log_level(void);
int dsv_lb2(unsigned n);
int dsv_yuv_write(FILE *out, int fno, DSV_PLANE *p);
int dsv_yuv_write_seq(FILE *out, DSV_PLANE *p);
int dsv_yuv_read(FILE *in, int fno, uint8_t *o, int w, int h, int subsamp);
int dsv_yuv_read_seq(FILE *in, uint8_t *o, int w, int h, int subsamp);
void dsv_post_process(DSV_PLANE *dp); /* decoder-side sharpen, dsv_internal.h:147 */
extern char *dsv_lvlname[5];

/* ------------------------------------------------------------------------- */
/* Section 5: hot-path seam, HOST buffers in / out (kernels run on the GPU)  */

void dsv_fwd_sbt(DSV_PLANE *src, DSV_COEFS *dst, DSV_FMETA *fm);           /* sbt.c:848 */
void dsv_inv_sbt(DSV_PLANE *dst, DSV_COEFS *src, int q, DSV_FMETA *fm);    /* sbt.c:890 */
void dsv_encode_plane(DSV_BS *bs, DSV_COEFS *src, int q, DSV_FMETA *fm);   /* hzcc.c:586 */
int dsv_decode_plane(DSV_BS *bs, DSV_COEFS *dst, int q, DSV_FMETA *fm);    /* hzcc.c:617 */
void dsv_sub_pred(DSV_MV *mv, DSV_PARAMS *p, DSV_FRAME *pred, DSV_FRAME *resd, DSV_FRAME *ref); /* bmc.c:1058 */
void dsv_add_res(DSV_MV *mv, DSV_FMETA *fm, int q, DSV_FRAME *resd, DSV_FRAME *pred, int do_filter); /* bmc.c:1073 */
void dsv_add_pred(DSV_MV *mv, DSV_FMETA *fm, int q, DSV_FRAME *resd, DSV_FRAME *out, DSV_FRAME *ref,
                  int do_filter);                                          /* bmc.c:1094 */
void dsv_intra_filter(int q, DSV_PARAMS *p, DSV_FMETA *fm, int c, DSV_PLANE *dp, int do_filter); /* bmc.c:391 */
DSV_MV *dsv_intra_analysis(DSV_FRAME *src, DSV_PARAMS *params);            /* hme.c:1836 */
void dsv_frame_copy(DSV_FRAME *dst, DSV_FRAME *src);                       /* frame.c:186 */
void dsv_ds2x_frame_luma(DSV_FRAME *dst, DSV_FRAME *src);                  /* frame.c:211 */
DSV_FRAME *dsv_extend_frame(DSV_FRAME *frame);                             /* frame.c:423 */
DSV_FRAME *dsv_extend_frame_luma(DSV_FRAME *frame);                        /* frame.c:413 */

/* reference dsv_encoder.h:202-215; src/ref/ogr[0] are full-size frames, [1..levels] the pyramid */
typedef struct {
    DSV_PARAMS *params;
    DSV_FRAME *src[DSV_MAX_PYRAMID_LEVELS + 1];
    DSV_FRAME *ref[DSV_MAX_PYRAMID_LEVELS + 1];
    DSV_FRAME *ogr[DSV_MAX_PYRAMID_LEVELS + 1];
    DSV_MV *mvf[DSV_MAX_PYRAMID_LEVELS + 1];
    DSV_MV *ref_mvf;
    DSV_MV mv_bank[128];
    int n_mv_bank_used;
    DSV_ENCODER *enc;
    int quant;
} DSV_HME;
int dsv_hme(DSV_HME *hme, int *scene_change_blocks, int *avg_err);         /* hme.c:2001 */

/* ------------------------------------------------------------------------- */
/* Section 6: library-specific entry points                                  */

/* 0 when a gfx950 device is usable, else a negative code; never falls back to the CPU */
int dsv2hip_device_ok(void);
const char *dsv2hip_version(void);
/* select the HIP device used by contexts created afterwards on this thread (one process per GPU: LOCAL_RANK) */
int dsv2hip_set_device(int ordinal);

/* dsv_enc for a packed planar 8-bit picture (Y, U, V back to back, no padding) that already
 * lives in device memory: the picture is not copied from the host.  Used by bench.py so that
 * the timed region starts with inputs resident in HBM. */
int dsv2hip_enc_device_frame(DSV_ENCODER *enc, const void *dev_planar, DSV_BUF *bufs);
/* lockstep step over n independent encoders of identical geometry: one frame each, the latency-bound
 * kernels (motion-estimation fronts, MC, in-loop filters) are launched once for all streams.  bufs has
 * 4 slots per stream; nbufs[k] = packets of stream k.  Output is identical to n separate dsv_enc calls. */
int dsv2hip_enc_batch(int n, DSV_ENCODER **encs, const void *const *dev_planar, DSV_BUF *bufs, int *nbufs);
/* the same step with the pictures in HOST memory, as dsv_enc (dsv_encoder.c:1430) receives them: host_planar[k] is
 * stream k's packed planar picture of this step.  host_next (NULL, or NULL entries, allowed) names the picture each
 * stream will bring to the NEXT call: it is uploaded on a copy stream under this step's kernels, and the next call
 * finds it in HBM when it passes the same pointer as host_planar[k] (the bytes must not change in between).  With
 * pictures in pinned memory (dsv2hip_host_alloc) every upload is asynchronous.  This is the entry point bench.py
 * times: the host-to-device transfer of every frame is inside the measured region. */
int dsv2hip_enc_batch_host(int n, DSV_ENCODER **encs, const void *const *host_planar, const void *const *host_next,
                           DSV_BUF *bufs, int *nbufs);
/* frame ingest / egress on the GPU.  (a) Packed pictures given to this encoder are interleaved UYVY
 * 4:2:2 rows: the de-interleave that dsv_yuv_read does on the host (dsv.c:177-205) happens in the ingest kernel;
 * the metadata must say DSV_SUBSAMP_UYVY or DSV_SUBSAMP_422.  (b) Every picture this decoder returns is delivered
 * as 4:2:0: the chroma conversions of the reference CLI's -out420p (util.c:79-153: conv444to422 + conv422to420,
 * conv422to420, conv411to420, conv410to420; dsv_main.c:1030-1048) run on the GPU as the picture is written to the
 * output frame.  Both return 0, or -1 when the stream's format does not allow it. */
int dsv2hip_enc_set_uyvy_input(DSV_ENCODER *enc, int on);
int dsv2hip_dec_set_out420p(DSV_DECODER *dec, int on);
void *dsv2hip_host_alloc(size_t bytes); /* pinned host memory (NULL on failure) */
void dsv2hip_host_free(void *p);
/* lockstep decode over n independent decoder instances: packet bufs[k] goes to decs[k]; ret[k], out[k]
 * and fn[k] are exactly what dsv_dec(decs[k], &bufs[k], &out[k], &fn[k]) would have produced (packets are
 * consumed the same way).  All pictures of a step run through one set of kernel launches. */
int dsv2hip_dec_batch(int n, DSV_DECODER **decs, DSV_BUF *bufs, DSV_FRAME **out, DSV_FNUM *fn, int *ret);
/* where the decoder parses the plane sections of the pictures it is given from now on (hzcc.c:451-585): 0 on the host (fastest with
 * ~16 host cores per GPU), 1 P pictures on the device (one wavefront per section: ~1.4 host cores per GPU), 2 every picture on the
 * device; -1 (the default when DSV2_DEC_DEVICE_PARSE is unset): by the number of cores the process may use.  Returns the mode in force. */
int dsv2hip_dec_parse_mode(void);
int dsv2hip_dec_set_parse_mode(int mode); /* 0 / 1 / 2 as above, < 0: back to the default; for pictures handed over after the call */
/* Submit queue behind dsv_enc / dsv_dec.  The reference's interface is one synchronous call per frame
 * (dsv_encoder.h:190-199, dsv_decoder.h:54-61); its own parallel recipe is one encoder per process
 * (parallel_encode_yuv.sh:31-52).  Threads that each loop dsv_enc (or dsv_dec) on an instance of their own are run
 * TOGETHER: calls that arrive within a bounded window (10 % of the last step, 100 us .. 2 ms; DSV2_COALESCE_US) and agree
 * on the picture geometry share one lockstep step, each call still returning when its own frame is finished, with the
 * packets dsv_enc alone would have produced.  DSV2_COALESCE=0 turns the queue off.  The stats calls report what it did:
 * out4[0] calls, [1] lockstep steps they ran as, [2] largest step, [3] microseconds leaders waited for expected callers. */
void dsv2hip_enc_queue_stats(unsigned long long *out4, int reset);
void dsv2hip_dec_queue_stats(unsigned long long *out4, int reset);
/* Encoder instances size their symbol (compaction) lists by need -- HALF the picture's coefficients (at least 65 536 symbols), the worst
 * case at once for lossless streams -- and enlarge them when a picture has more symbols (its symbols are then worked out a second time
 * and coded on the host; the packets are the same).  Number of such enlargements in this process so far: a stream pays at most
 * one.  DSV2_COMPACT_CAP=<symbols> overrides the initial size. */
long dsv2hip_enc_list_growths(void);
/* Device allocations that did not fit their instance's one-block arena (the block's size is an estimate): 0 unless the estimate has
 * drifted from the allocations it stands for. */
long dsv2hip_arena_fallbacks(void);
/* Test hook: the NEXT lockstep step of this process fails on purpose -- how = 1: as a motion search that did not deliver its
 * counters (the search has drained), how = 2: as a search token that never came (ingest / pyramids still enqueued).  The
 * failed step drains its streams before its job tables go back to the pool, releases the callers' frames, marks its encoders
 * dead; dsv_enc returns 0, the batch calls return -1 with every nbufs[k] = 0 (tests/test_gpu_robustness.py). */
void dsv2hip_test_fail_next_step(int how);
/* Residency census of a `make census` build (csrc/prio.h): resident wavefront-time of every kernel site, measured inside the
 * kernels while the lockstep groups share the chip.  dsv2hip_census_read writes one text line per site that ran -- "<file> <line>
 * <ticks of the 100 MHz clock x wavefronts> <workgroups> <wavefronts>" -- and returns the bytes written; the product build
 * carries no census and returns 0. */
void dsv2hip_census_reset(void);
int dsv2hip_census_read(char *out, int cap);
/* stage timing with HIP events on the stream each lockstep step runs on.  May be switched on and
 * off at any time (resets the totals).  dsv2hip_prof_read fills 9 entries (ingest+pyramid, HME,
 * predict, fwd SBT, quant+compact, inv SBT, reconstruct+filters, extend, and -- inside HME -- the level-0
 * search launch alone, the dominant kernel): milliseconds of stage span,
 * kernel launches, and *frames = steps folded in; dsv2hip_prof_read_units: stream-frames each
 * stage processed (what the algorithmic byte counts of DESIGN.md are multiplied by). */
void dsv2hip_prof_enable(int on);
int dsv2hip_prof_read(double *ms, long long *launches, long long *frames);
int dsv2hip_prof_read_units(long long *units);

/* Device-resident transform benchmark/ops handle: a plane set kept in HBM. */
typedef struct dsv2hip_planeset dsv2hip_planeset;
/* allocate device buffers for one picture (format, width, height) and its coefficient planes */
dsv2hip_planeset *dsv2hip_planeset_create(int format, int width, int height);
void dsv2hip_planeset_destroy(dsv2hip_planeset *ps);
/* copy a host frame (planar, any stride) into the device picture / back */
int dsv2hip_planeset_upload(dsv2hip_planeset *ps, const DSV_FRAME *frame);
int dsv2hip_planeset_download(dsv2hip_planeset *ps, DSV_FRAME *frame);
int dsv2hip_planeset_set_blockdata(dsv2hip_planeset *ps, const uint8_t *blockdata, int nblocks_h, int nblocks_v);
/* run the forward / inverse transform of plane c entirely in HBM (asynchronous on the set's stream) */
int dsv2hip_planeset_fwd_sbt(dsv2hip_planeset *ps, int c, int isP, int lossless);
int dsv2hip_planeset_inv_sbt(dsv2hip_planeset *ps, int c, int q, int isP, int lossless);
int dsv2hip_planeset_get_coefs(dsv2hip_planeset *ps, int c, DSV_SBC *out);
int dsv2hip_planeset_set_coefs(dsv2hip_planeset *ps, int c, const DSV_SBC *in);
int dsv2hip_planeset_sync(dsv2hip_planeset *ps);
/* times `iters` back-to-back launches of one transform with HIP events on the set's stream;
 * returns the mean milliseconds per call (negative on error) */
float dsv2hip_planeset_time_sbt(dsv2hip_planeset *ps, int c, int isP, int lossless, int inverse, int q, int iters);

#ifdef __cplusplus
}
#endif
#endif
