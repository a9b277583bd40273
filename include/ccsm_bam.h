/* libccsm_bam — native BGZF/BAM side of the call_mods path (host only: C++17, zlib, std::thread; no HIP, no htslib).
 *
 * What it replaces in the reference (PengNi/ccsmeth v0.5.0, through pysam/htslib):
 *   reader   extract_features.py:129-177 (worker_read_split_holebatches_to_queue: pysam.AlignmentFile(check_sq=False),
 *            get_forward_sequence(), tags fi/ri/fp/rp/fn/rn)        -> ccsm_bam_open / ccsm_bam_next
 *   writer   call_modifications.py:437-462 (_worker_write_modbam) + _bam2modbam.py:187-226 (_convert_locs_to_mmtag,
 *            _convert_probs_to_mltag, _refill_tags)                  -> ccsm_bam_writer_open / ccsm_bam_write_batch
 * A batch hands over the reads' arrays in exactly the layout ccsm_forward_reads_host (ccsm.h) takes, so a chunk of reads goes
 * file -> GPU -> file without a per-record Python object.  BGZF blocks are inflated / deflated by a pool of threads; record
 * order is preserved (the reference's --no_sort output).  Plain pointers and sizes; int status (0 = ok), text from
 * ccsm_bam_last_error().
 */
#ifndef CCSM_BAM_H_
#define CCSM_BAM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ccsm_bam_reader ccsm_bam_reader;
typedef struct ccsm_bam_writer ccsm_bam_writer;

/* One chunk of consecutive records.  Every pointer is owned by the batch (ccsm_bam_batch_free). */
typedef struct ccsm_bam_batch {
    int32_t n_reads;
    const uint8_t* records;     /* the raw BAM records back to back, each with its 4-byte block_size prefix */
    const int64_t* rec_offset;  /* (n_reads + 1) byte offsets into `records` */
    const int32_t* flag;        /* (n_reads) BAM FLAG */
    /* Read-level arrays in the layout of ccsm_reads (ccsm.h).  A read is "usable" when fi, ri, fp and rp are B:C arrays as
     * long as the sequence (extract_features.py:320-325 skips the others); unusable reads have length 0 here. */
    const int64_t* offset;      /* (n_reads) start of the read's bases in seq / fi / ri / fp / rp */
    const int32_t* length;      /* (n_reads) bases, 0 = unusable */
    const int32_t* n_sites;     /* (n_reads) CG sites that keep a full 21-mer window on both strands (extract_features.py:343-350) */
    const uint8_t* seq;         /* forward sequence, upper-case ASCII (reverse-complemented back for FLAG 0x10 records) */
    const uint8_t* fi;
    const uint8_t* ri;
    const uint8_t* fp;
    const uint8_t* rp;
    const float* fn;            /* (n_reads) */
    const float* rn;
    const uint64_t* name_hash;  /* (n_reads) 64-bit FNV-1a of the read name: the read's key for device-drawn initial states (ccsm_reads.h0_key) */
    int64_t total_bases;        /* bytes used in seq / fi / ri / fp / rp */
    uint64_t voffset_start;     /* BGZF virtual file offset (block file offset << 16 | offset in the block) of the first record */
    uint64_t voffset_end;       /* ... and of the byte behind the last one */
} ccsm_bam_batch;

const char* ccsm_bam_last_error(void);

/* threads = BGZF inflate workers (>= 1). */
int ccsm_bam_open(const char* path, int threads, ccsm_bam_reader** out);
/* SAM header text (not NUL-terminated: *text_len bytes) and the binary reference list exactly as stored (n_ref entries of
 * l_name, name, l_ref), for copying into the output file. */
int ccsm_bam_header(const ccsm_bam_reader* r, const char** text, int64_t* text_len, const uint8_t** refs, int64_t* refs_len,
                    int32_t* n_ref);
/* Next chunk of up to max_reads records (fewer at end of file; *out = NULL when no record is left). */
int ccsm_bam_next(ccsm_bam_reader* r, int32_t max_reads, ccsm_bam_batch** out);
void ccsm_bam_batch_free(ccsm_bam_batch* b);
void ccsm_bam_close(ccsm_bam_reader* r);
/* Random access for the multi-GPU call_mods (one scanning rank publishes the virtual offsets of the hole-batches, the others seek
 * instead of inflating the whole file): continue reading at voffset_start; with voffset_end != 0 no BGZF block behind the one
 * holding voffset_end is read or inflated (the range then ends like a file), so a rank inflates what it processes plus at most
 * one block per range.  ccsm_bam_tell = virtual offset of the next record; ccsm_bam_inflated_bytes = bytes inflated so far. */
int ccsm_bam_seek(ccsm_bam_reader* r, uint64_t voffset_start, uint64_t voffset_end);
int ccsm_bam_tell(ccsm_bam_reader* r, uint64_t* voffset);
/* Chunked reading without a scan (multi-GPU call_mods: every rank claims the next chunk number from a shared counter; nothing of the
 * file is inflated twice).  A chunk = the records that START in a BGZF block whose file offset lies in [coffset_lo, coffset_hi); the
 * last of them may end in a later chunk's blocks, which are then read as far as that record needs.  ccsm_bam_seek_chunk finds the
 * first BGZF block at or behind coffset_lo (gzip member header with the BC subfield, chained three deep), inflates from there and
 * locates the first record start by validating candidates (fixed fields in range for this header's reference count, NUL-terminated
 * printable name, CIGAR operations, auxiliary fields that walk to the exact end of the record, and a second valid record behind it);
 * subsequent ccsm_bam_next calls return the chunk's records and then NULL.  *voffset_first = virtual offset of that first record,
 * 0 when no record starts in the chunk.  The detection is a heuristic; what makes it safe is the caller's hand-over check: the
 * ccsm_bam_tell after a chunk's last batch must equal the next non-empty chunk's *voffset_first (call_mods raises otherwise). */
int ccsm_bam_seek_chunk(ccsm_bam_reader* r, uint64_t coffset_lo, uint64_t coffset_hi, uint64_t* voffset_first);
int64_t ccsm_bam_inflated_bytes(const ccsm_bam_reader* r);
/* What ccsm_bam_tell reports behind the LAST record of the file: (file offset of the last non-empty BGZF block << 16) | its ISIZE, from the
 * block headers and trailers alone (nothing is inflated, the reader's position does not move).  The hand-over chain of a chunked run must
 * END here: a chain that ends earlier has lost the tail of the input (a truncated last record, a last chunk whose record search ran off
 * the end of the file) - the error the reference's single reader raises from pysam (extract_features.py:129-177) instead of writing a
 * short modbam. */
int ccsm_bam_eof_voffset(ccsm_bam_reader* r, uint64_t* voffset);

/* level = zlib level of the BGZF blocks (1..9), threads = deflate workers. */
int ccsm_bam_writer_open(const char* path, const char* header_text, int64_t text_len, const uint8_t* refs, int64_t refs_len,
                         int32_t n_ref, int threads, int level, ccsm_bam_writer** out);
/* Writes the batch's records in order.  Old MM / ML tags are dropped, and fi / fp / ri / rp too when rm_pulse
 * (_bam2modbam.py:211-226).  Read r with tagged[r] != 0 gets  MM:Z:C+m?,<deltas>;  and  ML:B:C  built from its sites
 * k in [first_site[r], first_site[r+1]):  locs[k] = 0-based position of the called C in the forward sequence (ascending),
 * prob1[k] = the methylation probability as the reference rounds it (call_modifications.py:223).  A read whose
 * locations do not all sit on C's is written untagged, as the reference's failed assertion does (_bam2modbam.py:187-203).
 * *n_tagged = reads that received MM/ML. */
int ccsm_bam_write_batch(ccsm_bam_writer* w, const ccsm_bam_batch* b, const int32_t* first_site, const int32_t* locs,
                         const float* prob1, const uint8_t* tagged, int rm_pulse, int32_t* n_tagged);
/* Ends the current BGZF block: everything written so far is compressed and on disk, the next record starts a new block.
 * *file_offset = size of the file so far, so that [previous offset, *file_offset) is a self-contained run of BGZF blocks
 * (the multi-GPU call_mods stitches the per-rank files back into input order out of such runs). */
int ccsm_bam_writer_flush(ccsm_bam_writer* w, int64_t* file_offset);
int ccsm_bam_writer_close(ccsm_bam_writer* w);

/* ---- the index from the writer's own bookkeeping: no second pass over the output -------------------------------------------------
 * The reference indexes its modbam by re-reading it (pysam.index, call_modifications.py:602-606).  The writer knows where every record
 * lands: with tracking on it keeps, per placed record (reference id >= 0), the reference span and the virtual offsets of its first
 * byte and of the byte behind it, and per run (the blocks between two ccsm_bam_writer_flush calls) the record counts, whether the run
 * is in samtools' coordinate order, and the sort keys of its first and last record.  ccsm_bam_index_write turns a list of such run
 * tables - given in FINAL file order, each with the distance its blocks were moved by when the per-GPU part files were stitched -
 * into the .bai, or reports that the records are not in coordinate order (then no index is written and the caller sorts). */
typedef struct ccsm_bam_index_entry {
    int32_t tid, pos, end;     /* reference id, [pos, end) on the reference (one base for an unmapped record placed with its mate) */
    uint32_t flag;             /* BAM FLAG */
    uint64_t vbeg, vend;       /* virtual file offsets in the file the writer wrote */
} ccsm_bam_index_entry;
typedef struct ccsm_bam_index_run {
    int64_t n_records, n_unplaced;          /* records of the run(s); those without a reference id */
    int32_t sorted;                         /* 1: in coordinate order within */
    uint64_t first_k1; uint32_t first_k2;   /* sort key (reference id with unplaced last, position + 1 | reverse strand) of the first ... */
    uint64_t last_k1; uint32_t last_k2;     /* ... and the last record */
    int64_t n_entries;
    const ccsm_bam_index_entry* entries;    /* placed records in file order; owned by the writer until its next take / close */
    int64_t file_start, file_end;           /* byte span of the run(s) in the writer's file */
} ccsm_bam_index_run;
/* Switch tracking on / off; only at a run boundary (right after open + flush, or after any flush). */
int ccsm_bam_writer_track_index(ccsm_bam_writer* w, int enable);
/* The table of everything written since the previous take (one or more whole runs); must directly follow a flush. */
int ccsm_bam_writer_take_index(ccsm_bam_writer* w, ccsm_bam_index_run* out);
/* shift[i] (may be NULL = zeros) = final file offset of run i's first byte minus runs[i].file_start.  *sorted = 0: not in coordinate
 * order, nothing written. */
int ccsm_bam_index_write(const char* bai_path, int32_t n_ref, int32_t n_runs, const ccsm_bam_index_run* runs, const int64_t* shift,
                         int* sorted, int64_t* n_records);

/* ---- per-read modification calls of an aligned modbam projected on the reference: the feed of `ccsmeth call_freqb` -------
 * Replaces the per-read part of _readmods_to_bed_of_one_region (call_mods_freq_bam.py:486-537): record filters (:489-500),
 * haplotype tag (:502-508), MM/ML -> {query position: ML} (_get_moddict :179-205 / _get_moddict_in_tags :125-175) and the walk
 * over get_aligned_pairs (:509-536).  One output row per (record, reference position) the reference would append to its
 * refposinfo / refposinfo_rev dictionaries; rows of a record are in CIGAR order, records in batch order.  The caller groups
 * the rows by region and position. */
typedef struct ccsm_bam_modcall_opts {
    double identity;           /* records with (M + =) / (M+I+D+N+P+=+X+B) < identity are skipped (process_utils.py:174-186) */
    int32_t mapq;              /* records with MAPQ < mapq are skipped */
    int32_t no_supplementary;  /* skip FLAG 0x800 */
    int32_t base_clip;         /* drop this many aligned pairs at either end of the record (:512-513) */
    int32_t refsites_all;      /* walk every CIGAR column (matches_only=False) and report ML 0 at reference motif sites of the
                                * record's strand that carry no call (site_mask must be given) */
    char hap_tag[2];           /* "HP" */
    char modbase;              /* 'C' */
    char modification;         /* 'm' */
} ccsm_bam_modcall_opts;

typedef struct ccsm_bam_modcalls {
    int64_t n;                 /* rows */
    const int32_t* tid;        /* reference id of the record */
    const int32_t* pos;        /* 0-based reference position */
    const uint8_t* strand;     /* 0 = forward record, 1 = reverse (FLAG 0x10) */
    const uint8_t* ml;         /* ML byte; probability = round(ml/256 + 1e-6, 6), 0 for ml == 0 (_cal_mod_prob :102-107) */
    const uint8_t* hap;        /* 1, 2, or 0 for anything else / no tag */
    int64_t n_records;         /* records looked at */
    int64_t n_used;            /* records that passed the filters */
} ccsm_bam_modcalls;

/* site_mask[tid] (may be NULL per contig, or site_mask itself NULL): one byte per reference base, bit 0 = motif site on the
 * forward strand, bit 1 = on the reverse strand; mask_len[tid] = bytes.  threads = workers over the batch's records. */
int ccsm_bam_modcalls_of_batch(const ccsm_bam_batch* b, const ccsm_bam_modcall_opts* opts, const uint8_t* const* site_mask,
                               const int64_t* mask_len, int32_t n_ref, int threads, ccsm_bam_modcalls** out);
void ccsm_bam_modcalls_free(ccsm_bam_modcalls* c);

/* ---- post-processing of the output modbam: the reference's pysam.sort + pysam.index (call_modifications.py:592-607) ----------
 * ccsm_bam_index_build streams the file once: *sorted = 1 when the records are in samtools' coordinate order (reference id with
 * unmapped last, position, forward before reverse strand) and then writes the BAI index (SAM spec 5.2: binning index, 16 kb
 * linear index, htslib's metadata pseudo-bin 37450, n_no_coor); *sorted = 0 leaves no index.  ccsm_bam_sort rewrites the file
 * in that order (stable, in memory: fails when the uncompressed records exceed max_bytes > 0) with @HD SO:coordinate. */
int ccsm_bam_index_build(const char* bam_path, const char* bai_path, int threads, int* sorted, int64_t* n_records);
int ccsm_bam_sort(const char* in_path, const char* out_path, int threads, int level, int64_t max_bytes);

/* Per record of a batch, what --mode align of extract_features.py needs besides FLAG (extract_features.py:88-126, 272-304):
 * MAPQ, pysam's query_alignment_start / query_alignment_end (soft clips excluded, hard clips are not in SEQ) and the CIGAR
 * identity (M + =) / (M+I+D+N+P+=+X+B) of process_utils.py:174-186 (0 for an empty CIGAR). */
int ccsm_bam_align_info(const ccsm_bam_batch* b, int32_t* mapq, int32_t* qstart, int32_t* qend, double* identity);

#ifdef __cplusplus
}
#endif
#endif
