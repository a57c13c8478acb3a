/*
 * gorder_xtc.h — C ABI of the host-side XTC trajectory reader.
 *
 * Replaces, for the drop-in path, what gorder reaches through groan_rs' `GroupXtcReader`
 * (call sites /root/reference/src/analysis/common.rs:281-304): sequential decode of GROMACS
 * xdr3dcoord-compressed frames, conversion of only the atoms of the "Master" group, the time window
 * `begin/end/step` (common.rs:239-246) and concatenation of several files with the duplicate
 * boundary frame dropped (CHANGELOG.md:64).
 *
 * The third-party decoder (molly 0.5.0 via groan_rs 0.11.2) is not in /root/reference; the wire
 * format restated here is the published GROMACS XTC format (magic 1995, "magic ints" mixed-radix
 * packing).  int -> f32 conversion: coordinate = int * (1 / precision) as in GROMACS' xdrfile.
 */
#ifndef GORDER_XTC_H
#define GORDER_XTC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gorder_xtc_reader gorder_xtc_reader;

typedef enum {
    GORDER_XTC_OK = 0,
    GORDER_XTC_EOF = 1,            /* clean end of file */
    GORDER_XTC_ERR_OPEN = -1,
    GORDER_XTC_ERR_FORMAT = -2,    /* bad magic / truncated frame / corrupt bit stream */
    GORDER_XTC_ERR_ARGUMENT = -3,
    GORDER_XTC_ERR_NO_SPACE = -4   /* gorder_xtc_pack_window*: the next selected frame does not fit what is left of the blob;
                                    * reader, *state and *last_time are as they were before the call */
} gorder_xtc_status_t;

/* Open one file: XTC, or — recognised by its magic number 1993 — a GROMACS TRR file (uncompressed single- or
 * double-precision positions; frames without positions are skipped; `precision` is reported as 0), which the
 * reference reads through groan_rs' TrrReader (common.rs:306-320); a file with neither magic number is read as a
 * multi-frame GRO text trajectory (GroReader, common.rs:322-333: time from "t=" in the title line, positions in
 * three equal-width columns from column 20, 3- or 9-value box line).  `group` (may be NULL = all atoms) lists the n_group atom indices to convert;
 * decoded frames hold exactly those atoms, in that order (the "Master" group of common.rs:283-304).  The compressed
 * stream of a frame is only decompressed up to the last atom of the group: a group of lipids in front of the water
 * costs the lipids' share of the frame. */
int gorder_xtc_open(const char *path, const uint32_t *group, uint32_t n_group, gorder_xtc_reader **out);
void gorder_xtc_close(gorder_xtc_reader *r);
uint32_t gorder_xtc_n_atoms_file(const gorder_xtc_reader *r);   /* atoms per frame in the file */
uint32_t gorder_xtc_n_atoms_out(const gorder_xtc_reader *r);    /* atoms per decoded frame */

/* Decode the next frame: xyz [n_atoms_out][3] nm, box [3][3], step, time (ps), precision.
 * Pass xyz = NULL to skip the coordinate block (header only; cheap). */
int gorder_xtc_next(gorder_xtc_reader *r, float *xyz, float *box9, int64_t *step, float *time_ps,
                    float *precision);

/* Decode up to `capacity` frames selected by the time window into caller buffers
 * xyz [capacity][n_atoms_out][3], box [capacity][3][3], time [capacity]:
 *   begin/end in ps (negative end = no limit), every `step`-th frame of the window.
 * `*state` (init 0) carries the frame counter across calls/files; `*last_time` (init -inf) the time of
 * the previous analysed-or-skipped frame so that a frame whose time equals the last frame of the
 * previous file is dropped when reading concatenated trajectories.
 * Returns the number of frames written (0 at end of file) or a negative gorder_xtc_status_t. */
int64_t gorder_xtc_read_window(gorder_xtc_reader *r, float begin_ps, float end_ps, uint32_t step,
                               uint64_t *state, double *last_time, float *xyz, float *box9, float *time_ps,
                               uint64_t capacity);

/* Pass over up to `max_frames` frames the window selects WITHOUT decoding them (headers only; any format): same
 * selection, `state` and `last_time` bookkeeping as gorder_xtc_read_window.  Returns how many were passed over (fewer
 * than max_frames: the file ended) or a negative status.  With max_frames = UINT64_MAX it counts the selected frames of
 * a file; with a smaller number it positions a reader at the start of a rank's shard of the trajectory. */
int64_t gorder_xtc_skip_window(gorder_xtc_reader *r, float begin_ps, float end_ps, uint32_t step, uint64_t *state,
                               double *last_time, uint64_t max_frames);

/* The same with `n_threads` decoder threads: the frame headers are scanned sequentially (cheap), the selected
 * frames are then decompressed in parallel, each worker through a file handle of its own.  Output, state and
 * return value are identical to gorder_xtc_read_window for every n_threads (the reference decodes one reader
 * thread per analysis thread, common.rs:283-339; decoding is its bottleneck). */
int64_t gorder_xtc_read_window_mt(gorder_xtc_reader *r, float begin_ps, float end_ps, uint32_t step,
                                  uint64_t *state, double *last_time, float *xyz, float *box9, float *time_ps,
                                  uint64_t capacity, uint32_t n_threads);

/* ---- frames for the device-side decoder (gorder_hip_xtc_decode, include/gorder_hip.h) -----------------------------
 * Decoding the bit stream is the dominant cost of reading a trajectory ("optional GPU bit-unpack
 * later").  Instead of decoding, gorder_xtc_pack_window copies the still-compressed coordinate blocks of the selected
 * frames into one caller buffer (`blob`, meant to be pinned host memory) and describes each with a gorder_xtc_frame_t;
 * the device decodes them (a wave per frame finds where chunks of 256 atoms start, a lane per chunk unpacks).  Frame selection (time window, step, duplicate boundary frame), `state`,
 * `last_time`, box and time outputs are exactly those of gorder_xtc_read_window.  XTC only (not TRR / GRO). */
typedef struct {
    uint64_t offset;         /* of the frame's bit stream in the blob: a multiple of 64; the stream is followed by zeros up to
                                the next multiple of 64 and by 64 more (the device reads whole 64-byte pieces) */
    uint64_t recip1, recip2; /* floor(2^64 / sizeint[1]), floor(2^64 / sizeint[2]) (all ones for a size of 1) */
    uint32_t n_bytes;        /* length of the bit stream, padded to a multiple of 4 as in the file */
    uint32_t kind;           /* bit 0: raw big-endian floats (files of <= 9 atoms) instead of a compressed block;
                                bit 1: only a leading part of the block was copied (gorder_xtc_pack_window_ex) */
    int32_t minint[3];
    uint32_t sizeint[3];     /* maxint - minint + 1 */
    int32_t smallidx;
    float inv_precision;     /* 1 / precision, in f32 like gorder_xtc_next */
    uint32_t bitsize;        /* width of a full atom as ONE mixed-radix number; 0: three fields of bitsizeint bits */
    uint32_t bitsizeint;     /* the three field widths, 8 bits each (x lowest) */
} gorder_xtc_frame_t;

/* Returns the number of frames packed (0 at the end of the file), fewer than `capacity` also when the next frame
 * would not fit `blob_capacity` (a later call continues with that frame), or a negative gorder_xtc_status_t
 * (GORDER_XTC_ERR_NO_SPACE when not even ONE frame fits the blob — nothing has happened then: file position, *state
 * and *last_time are those of the call's entry; GORDER_XTC_ERR_ARGUMENT for a TRR / GRO reader; GORDER_XTC_ERR_FORMAT
 * also for a block whose byte count runs past the end of the file).
 * `*blob_bytes` receives the bytes of the blob in use.  The blocks are copied by `n_threads` threads: out of a read-only
 * mapping of the file made at the reader's first window (streaming stores into the blob), `pread` where there is no
 * mapping.  A file that has shrunk since it was mapped is read by `pread` from then on, the part of a file that has
 * grown behind the mapping too.  What a mapping cannot turn into an error code is a file truncated, or an I/O error of the
 * file system, WHILE a copy is reading those pages: that is SIGBUS.  For trajectories a running simulation rewrites, and
 * on network file systems, set GORDER_XTC_PREAD=1 (every byte by pread, errors as GORDER_XTC_ERR_FORMAT). */
int64_t gorder_xtc_pack_window(gorder_xtc_reader *r, float begin_ps, float end_ps, uint32_t step, uint64_t *state,
                               double *last_time, uint8_t *blob, uint64_t blob_capacity, uint64_t *blob_bytes,
                               gorder_xtc_frame_t *frames, float *box9, float *time_ps, uint64_t capacity,
                               uint32_t n_threads);
/* The same with the block copies handed to a pool of copying threads: the call returns when the headers are scanned
 * (frames, boxes, times and *blob_bytes are final), the blob's bytes are complete once gorder_xtc_pool_wait has
 * returned (its status: the first copy failure since the last wait).  The sequential header scan of the next file
 * or window then overlaps these copies.  The reader may be closed before the wait. */
typedef struct gorder_xtc_pool gorder_xtc_pool;
int gorder_xtc_pool_create(uint32_t n_threads, gorder_xtc_pool **out);
int gorder_xtc_pool_wait(gorder_xtc_pool *pool);
void gorder_xtc_pool_destroy(gorder_xtc_pool *pool);
int64_t gorder_xtc_pack_window_pool(gorder_xtc_reader *r, float begin_ps, float end_ps, uint32_t step, uint64_t *state,
                                    double *last_time, uint8_t *blob, uint64_t blob_capacity, uint64_t *blob_bytes,
                                    gorder_xtc_frame_t *frames, float *box9, float *time_ps, uint64_t capacity,
                                    gorder_xtc_pool *pool);

/* The general form.  `pool` may be NULL (then `n_threads` copying threads, synchronously).  `prefix_q16` < 65536 copies
 * only the leading prefix_q16 / 65536 of every block (+ 2 KB): the analysed atoms come first in a frame and the decoder
 * stops behind them, so the tail — the solvent — need not travel; such a frame has bit 1 of `kind` set and `n_bytes` =
 * the bytes copied, and a decoder that runs past them must report the frame as SHORT, not as corrupt (k_xtc_scan
 * does; the trajectory driver then decodes that frame on the host from `file_pos`).  `file_pos` (may be NULL)
 * receives the file offset of every packed frame's header, for gorder_xtc_read_at. */
int64_t gorder_xtc_pack_window_ex(gorder_xtc_reader *r, float begin_ps, float end_ps, uint32_t step, uint64_t *state,
                                  double *last_time, uint8_t *blob, uint64_t blob_capacity, uint64_t *blob_bytes,
                                  gorder_xtc_frame_t *frames, float *box9, float *time_ps, uint64_t capacity,
                                  uint32_t n_threads, gorder_xtc_pool *pool, uint32_t prefix_q16, int64_t *file_pos);
/* Decode the frame whose header starts at `file_pos` (as reported by gorder_xtc_pack_window_ex); moves the reader. */
int gorder_xtc_read_at(gorder_xtc_reader *r, int64_t file_pos, float *xyz, float *box9);

/* Look at a file's first bytes only: 1 = XTC (and *n_atoms = atoms per frame, *file_bytes = size of the file,
 * *first_frame_bytes = header + coordinate block of its first frame; each may be NULL), 0 = something else
 * (TRR, GRO, ...), negative = cannot be opened / too short. */
int gorder_xtc_probe(const char *path, uint32_t *n_atoms, uint64_t *file_bytes, uint32_t *first_frame_bytes);
/* 1 when the reader's file is an XTC file (what gorder_xtc_pack_window accepts), else 0 */
int gorder_xtc_is_xtc(const gorder_xtc_reader *r);
/* atoms of a frame the decoder has to go through: up to the last atom of the group (all atoms without a group) */
uint32_t gorder_xtc_n_atoms_needed(const gorder_xtc_reader *r);

/* ---- writer (tooling) --------------------------------------------------------------------------------
 * The compression side of the same format, so that tests and the end-to-end benchmark can produce multi-frame
 * XTC input for the reader -> GPU pipeline (the reference itself never writes trajectories).  Frames of up to
 * 9 atoms are stored as raw floats, larger ones compressed with `precision` grid steps per nm (GROMACS default
 * 1000).  A frame written here and read back through gorder_xtc_next gives round(x * precision) / precision. */
typedef struct gorder_xtc_writer gorder_xtc_writer;
int gorder_xtc_writer_open(const char *path, uint32_t n_atoms, float precision, gorder_xtc_writer **out);
int gorder_xtc_writer_add(gorder_xtc_writer *w, const float *xyz, const float *box9, int64_t step, float time_ps);
void gorder_xtc_writer_close(gorder_xtc_writer *w);

#ifdef __cplusplus
}
#endif
#endif
