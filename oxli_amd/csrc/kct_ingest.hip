// kct_ingest.hip -- FASTA / FASTQ file ingestion (kct_consume_file): the caller side of the path.
#include "kct_internal.h"
#include "parallel_inflate.h"

#include <deque>
#include <emmintrin.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <functional>
#include <memory>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace kcth;

// ---- FASTA / FASTQ ingestion: the caller side of the path (README.md:89-99) ---------------------------
// The reference delegates parsing to screed and calls consume() once per record.  Here host parsers
// turn the file into record-stream chunks in pinned memory while ONE worker thread uploads and counts
// the finished chunks (the table is not thread-safe), so parsing and device work overlap.
//   * plain files are mapped and cut into segments that several parser threads take in turn; a thread
//     starts at the first record start at or after its segment's first byte and stops at the first record
//     start at or after the segment's end, so every record is parsed by exactly one thread.  Records are
//     independent, so the chunks can be counted in any order;
//   * a plain gzip stream is inflated by ONE thread of its own into a ring of text slots that the calling thread parses (zlib's
//     inflate, ~0.4 GB/s of text, is then the bound -- but no longer inflate PLUS parse);
//   * BGZF (bgzip's blocked gzip: every member announces its compressed size in an extra field and its text size in its trailer, so
//     blocks are found without inflating) is cut into slots of whole blocks that SEVERAL threads take in turn: a thread inflates
//     its slot (libdeflate when the system has it, else zlib) and parses the records that start AND end inside it; what lies before
//     the slot's first record start and from its last record start on goes, in slot order, to the calling thread, which parses
//     that stream of fragments -- one record per slot boundary, or all of it when records are longer than slots.
// A record longer than a chunk is cut with a (k-1)-base overlap, which keeps every window counted
// exactly once.
namespace {

struct FileChunk {
    PinnedBuf *host = nullptr;
    size_t used = 0;
    bool in_flight = false;
};

// Hands finished chunks to the worker thread and waits for buffers to come back.
struct ChunkQueue {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<FileChunk *> q;
    bool done = false;
    kct_status status = KCT_OK;  // first failure of the worker (or of a parser): everyone stops
    std::string msg;
    u64 counted = 0;

    void submit(FileChunk *c) {
        { std::lock_guard<std::mutex> lk(mu); c->in_flight = true; q.push_back(c); }
        cv.notify_all();
    }
    void wait_free(FileChunk *c) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !c->in_flight; });
    }
    bool failed() { std::lock_guard<std::mutex> lk(mu); return status != KCT_OK; }
    void fail(kct_status st, const char *m) {
        { std::lock_guard<std::mutex> lk(mu); if (status == KCT_OK) { status = st; msg = m; } }
        cv.notify_all();
    }
};

// Appends records to a pair of chunk buffers; a full buffer goes to the queue.
struct ChunkWriter {
    ChunkQueue *queue;
    FileChunk chunk[2];
    int cur = 0;
    size_t cap, full_cap, k;
    unsigned char *out;
    size_t used = 0, rec_len = 0;  // rec_len = bases of the current record already in this chunk
    u64 records = 0, bases = 0;

    // (a writer's FIRST chunk is cut at 2 MiB: the uploads start 0.3 ms into the call instead of 2.2 -- the link is the bound of a plain file)
    ChunkWriter(ChunkQueue *q, PinnedBuf *a, PinnedBuf *b, size_t chunk_cap, size_t ksize)
        : queue(q), cap(chunk_cap > ((size_t)8 << 20) ? chunk_cap : std::min<size_t>(chunk_cap, (size_t)2 << 20)), full_cap(chunk_cap), k(ksize) {   // (not a small table's 16 MiB chunks: each is a pass there)
        chunk[0].host = a; chunk[1].host = b;
        out = (unsigned char *)a->p;
    }
    void flush(bool mid_record) {
        unsigned char tail[256];  // the last k-1 bases of an unfinished record open the next chunk
        size_t ntail = 0;
        if (mid_record) { ntail = std::min(rec_len, k - 1); memcpy(tail, out + used - ntail, ntail); }
        chunk[cur].used = used;
        queue->submit(&chunk[cur]);
        cur ^= 1;
        cap = full_cap;
        queue->wait_free(&chunk[cur]);
        out = (unsigned char *)chunk[cur].host->p;
        memcpy(out, tail, ntail);
        used = ntail;
        rec_len = ntail;
    }
    void emit(const unsigned char *p, size_t n) {  // sequence bytes of the current record
        while (n) {
            if (used + 1 >= cap) flush(true);
            const size_t take = std::min(n, cap - 1 - used);
            memcpy(out + used, p, take);
            used += take; rec_len += take; p += take; n -= take; bases += take;
        }
    }
    void end_record() {
        if (used + 1 >= cap) flush(true);
        out[used++] = '\n';
        rec_len = 0;
        ++records;
    }
    void finish() {
        if (used) { chunk[cur].used = used; queue->submit(&chunk[cur]); used = 0; }
        queue->wait_free(&chunk[0]);
        queue->wait_free(&chunk[1]);
    }
};

// Byte sources: a ring of inflated text (gzip input), or a mapped file.
// ---- gzip input: inflater threads fill a ring of text slots in order, the parser reads them in order -------------------------------
struct TextRing {
    struct Slot { std::vector<unsigned char> buf; size_t used = 0; bool full = false; };
    std::vector<Slot> slots;
    std::mutex mu;
    std::condition_variable cv;
    u64 consumed = 0;      // slots the parser has finished with (slot seq is free once seq < consumed + slots.size())
    u64 end_seq = ~0ULL;   // sequence number one past the last slot (set by the producer that meets the end of the input)
    bool failed = false;
    std::string msg;
    TextRing(size_t n, size_t bytes) : slots(n) { for (auto &sl : slots) sl.buf.resize(bytes); }
    // producer: the slot for sequence number seq, once the parser has released it (nullptr: the job was abandoned)
    Slot *acquire(u64 seq) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return failed || seq < consumed + slots.size(); });
        return failed ? nullptr : &slots[seq % slots.size()];
    }
    void publish(Slot *sl, size_t used) {
        { std::lock_guard<std::mutex> lk(mu); sl->used = used; sl->full = true; }
        cv.notify_all();
    }
    void finish(u64 seq_end) {
        { std::lock_guard<std::mutex> lk(mu); end_seq = std::min(end_seq, seq_end); }
        cv.notify_all();
    }
    void fail(const char *m) {
        { std::lock_guard<std::mutex> lk(mu); if (!failed) { failed = true; msg = m; } }
        cv.notify_all();
    }
    // parser: the next full slot in order (nullptr at the end of the input or on failure)
    Slot *next(u64 seq) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return failed || seq >= end_seq || slots[seq % slots.size()].full; });
        if (failed || seq >= end_seq) return nullptr;
        return &slots[seq % slots.size()];
    }
    void release(u64 seq) {
        { std::lock_guard<std::mutex> lk(mu); slots[seq % slots.size()].full = false; consumed = seq + 1; }
        cv.notify_all();
    }
};

struct RingSource {
    TextRing *ring;
    TextRing::Slot *cur = nullptr;
    u64 seq = 0;
    size_t pos = 0;
    bool eof = false;
    bool fill() {
        if (eof) return false;
        if (cur) { ring->release(seq); ++seq; cur = nullptr; }
        for (;;) {
            cur = ring->next(seq);
            if (!cur) { eof = true; return false; }
            pos = 0;
            if (cur->used) return true;
            ring->release(seq); ++seq; cur = nullptr;   // (an empty slot: BGZF's end-of-file block)
        }
    }
    int peek() { if ((!cur || pos >= cur->used) && !fill()) return -1; return cur->buf[pos]; }
    bool span(const unsigned char *&b, size_t &avail) { if ((!cur || pos >= cur->used) && !fill()) return false; b = cur->buf.data() + pos; avail = cur->used - pos; return true; }
    void advance(size_t n) { pos += n; }
    size_t offset() const { return 0; }  // unused: one parser, no segment end
};

// BGZF (SAM specification 4.1): gzip members of at most 64 KiB with FEXTRA holding the subfield 'B' 'C' = (member size - 1)
struct BgzfBlock { size_t off, csize; unsigned isize; };
bool bgzf_scan(const unsigned char *p, size_t size, std::vector<BgzfBlock> &out) {
    size_t off = 0;
    while (off < size) {
        if (size - off < 28 || p[off] != 0x1f || p[off + 1] != 0x8b || p[off + 2] != 8 || !(p[off + 3] & 4)) return false;
        const size_t xlen = p[off + 10] | ((size_t)p[off + 11] << 8);
        size_t x = off + 12, bsize = 0;
        const size_t xend = x + xlen;
        if (xend > size) return false;
        while (x + 4 <= xend) {
            const size_t slen = p[x + 2] | ((size_t)p[x + 3] << 8);
            if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2 && x + 6 <= xend) bsize = (p[x + 4] | ((size_t)p[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (!bsize || off + bsize > size || bsize < 12 + xlen + 8 || (p[off + 3] & ~4)) return false;   // (other header flags: not bgzip's output)
        BgzfBlock b;
        b.off = off; b.csize = bsize;
        memcpy(&b.isize, p + off + bsize - 4, 4);
        if (b.isize > 65536) return false;
        out.push_back(b);
        off += bsize;
    }
    return !out.empty();
}

struct MemSource {
    const unsigned char *base, *p, *end;
    int peek() { return p < end ? *p : -1; }
    bool span(const unsigned char *&b, size_t &avail) { if (p >= end) return false; b = p; avail = (size_t)(end - p); return true; }
    void advance(size_t n) { p += n; }
    size_t offset() const { return (size_t)(p - base); }
};

template <class Src>
void skip_line(Src &s) {
    const unsigned char *b; size_t avail;
    while (s.span(b, avail)) {
        const unsigned char *nl = (const unsigned char *)memchr(b, '\n', avail);
        s.advance(nl ? (size_t)(nl - b) + 1 : avail);
        if (nl) return;
    }
}

// copies the rest of the current line (without CR/LF) into the record; returns its length
template <class Src>
size_t emit_line(Src &s, ChunkWriter &w) {
    size_t total = 0;
    const unsigned char *b; size_t avail;
    while (s.span(b, avail)) {
        const unsigned char *nl = (const unsigned char *)memchr(b, '\n', avail);
        const size_t n = nl ? (size_t)(nl - b) : avail;
        size_t m = n;
        if (m && b[m - 1] == '\r') --m;
        w.emit(b, m); total += m;
        s.advance(n + (nl ? 1 : 0));
        if (nl) break;
    }
    return total;
}

// no line break (nothing below ' ', in fact, and nothing from 0x80 on) among b[0, n): sixteen bytes at a time (SSE2, the x86-64 baseline)
inline bool plain_bytes(const unsigned char *b, size_t n) {
    const __m128i lim = _mm_set1_epi8(32);
    size_t i = 0;
    for (; i + 16 <= n; i += 16)
        if (_mm_movemask_epi8(_mm_cmplt_epi8(_mm_loadu_si128((const __m128i *)(b + i)), lim))) return false;
    if (i < n) {
        if (n >= 16) return _mm_movemask_epi8(_mm_cmplt_epi8(_mm_loadu_si128((const __m128i *)(b + n - 16)), lim)) == 0;
        for (; i < n; ++i) if (b[i] < 32 || b[i] >= 128) return false;
    }
    return true;
}

// Parses records until the source ends or (segmented sources) a record would start at or after `stop`.
// `fmt` is '>' or '@'.  Returns false (with set_err) on a malformed file.
template <class Src>
bool parse_records(Src &s, ChunkWriter &w, int fmt, size_t stop, ChunkQueue &queue, const char *path) {
    int c;
    u64 seen = 0;
    while ((c = s.peek()) >= 0) {
        if (c == '\n' || c == '\r' || c == ' ' || c == '\t') { s.advance(1); continue; }
        if (s.offset() >= stop) break;
        if (c != fmt) { set_err("%s: malformed record header near record %llu", path, (unsigned long long)w.records); return false; }
        skip_line(s);  // header
        size_t seq_len = 0;
        if (fmt == '>') {
            while ((c = s.peek()) >= 0 && c != '>') seq_len += emit_line(s, w);
        } else {
            while ((c = s.peek()) >= 0 && c != '+') seq_len += emit_line(s, w);
            skip_line(s);  // '+' line
            size_t q = 0;  // quality: as many characters as the sequence had
            const unsigned char *b; size_t avail;
            // (the usual record -- the quality string on ONE line, as long as the sequence -- is stepped over in one piece: counting its
            // characters one by one, as the loop below does for whatever else there is, was three quarters of a 150 bp record's parse)
            if (s.span(b, avail) && avail > seq_len && (b[seq_len] == '\n' || b[seq_len] == '\r') && plain_bytes(b, seq_len)) { s.advance(seq_len); q = seq_len; }
            while (q < seq_len && s.span(b, avail)) {
                size_t i = 0;
                for (; i < avail && q < seq_len; ++i) q += (b[i] != '\n' && b[i] != '\r');
                s.advance(i);
            }
            skip_line(s);
        }
        w.end_record();
        if ((++seen & 1023) == 0 && queue.failed()) return true;  // the worker failed: stop early, its error is reported
    }
    return true;
}

// Is the line that starts at `ls` a record header?  FASTQ: '@' may also open a quality line.  A header is followed by a sequence
// line and a '+' line; a quality line is followed by the next header and ITS sequence line, which never starts with '+'.
// (Undecidable -- the two following lines are not all there -- counts as no.)
inline bool is_record_start(const unsigned char *ls, const unsigned char *end, int fmt) {
    if (ls >= end || *ls != (unsigned char)fmt) return false;
    if (fmt == '>') return true;
    auto line_end = [&](const unsigned char *q) { const unsigned char *nl = (const unsigned char *)memchr(q, '\n', (size_t)(end - q)); return nl ? nl : end; };
    const unsigned char *l1 = line_end(ls);                       // end of the candidate header
    const unsigned char *l2 = l1 < end ? line_end(l1 + 1) : end;  // end of the sequence line
    return l2 < end && l2 + 1 < end && l2[1] == '+';
}

// First record start at or after `from` in a mapped file (file size if there is none).
size_t find_record_start(const unsigned char *base, size_t size, size_t from, int fmt) {
    if (from == 0) return 0;
    if (from >= size) return size;
    // line starts at or after `from`: one after every '\n' at or after from - 1
    const unsigned char *p = base + from - 1, *end = base + size;
    while (p < end) {
        const unsigned char *nl = (const unsigned char *)memchr(p, '\n', (size_t)(end - p));
        if (!nl || nl + 1 >= end) return size;
        const unsigned char *ls = nl + 1;
        if (is_record_start(ls, end, fmt)) return (size_t)(ls - base);
        p = ls;
    }
    return size;
}

// Last record start at or after `lo` (itself a record start) in base[0, size).
size_t find_last_record_start(const unsigned char *base, size_t size, size_t lo, int fmt) {
    const unsigned char *end = base + size, *q = end;
    while (q > base + lo) {
        const unsigned char *nl = (const unsigned char *)memrchr(base + lo, '\n', (size_t)(q - (base + lo)));
        if (!nl) break;
        if (is_record_start(nl + 1, end, fmt)) return (size_t)(nl + 1 - base);
        q = nl;
    }
    return lo;
}

// ---- BGZF: what the slot parsers leave for the stitching parser, in slot order --------------------------------------------------
struct Fragments {
    struct Piece { std::vector<unsigned char> head, tail; bool ready = false; };
    std::vector<Piece> ring;
    std::mutex mu;
    std::condition_variable cv;
    u64 consumed = 0, end_seq;
    bool failed = false;
    std::string msg;
    Fragments(size_t n, u64 nslots) : ring(n), end_seq(nslots) {}
    bool may_start(u64 seq) {   // a slot thread: not more than ring.size() slots ahead of the stitcher
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return failed || seq < consumed + ring.size(); });
        return !failed;
    }
    void publish(u64 seq, const unsigned char *h, size_t nh, const unsigned char *t, size_t nt) {
        {
            std::lock_guard<std::mutex> lk(mu);
            Piece &p = ring[seq % ring.size()];
            p.head.assign(h, h + nh); p.tail.assign(t, t + nt); p.ready = true;
        }
        cv.notify_all();
    }
    void fail(const char *m) {
        { std::lock_guard<std::mutex> lk(mu); if (!failed) { failed = true; msg = m; } }
        cv.notify_all();
    }
    Piece *next(u64 seq) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return failed || seq >= end_seq || ring[seq % ring.size()].ready; });
        return failed || seq >= end_seq ? nullptr : &ring[seq % ring.size()];
    }
    void release(u64 seq) {
        { std::lock_guard<std::mutex> lk(mu); ring[seq % ring.size()].ready = false; consumed = seq + 1; }
        cv.notify_all();
    }
};

struct FragmentSource {   // head 0, tail 0, head 1, tail 1, ... as one byte stream
    Fragments *f;
    Fragments::Piece *cur = nullptr;
    u64 seq = 0;
    int part = 0;
    size_t pos = 0;
    bool eof = false;
    const std::vector<unsigned char> &buf() const { return part ? cur->tail : cur->head; }
    bool fill() {
        for (;;) {
            if (eof) return false;
            if (!cur) {
                cur = f->next(seq);
                if (!cur) { eof = true; return false; }
                part = 0; pos = 0;
            }
            if (pos < buf().size()) return true;
            if (part == 0) { part = 1; pos = 0; continue; }
            f->release(seq); ++seq; cur = nullptr;
        }
    }
    int peek() { if ((!cur || pos >= buf().size()) && !fill()) return -1; return buf()[pos]; }
    bool span(const unsigned char *&b, size_t &avail) { if ((!cur || pos >= buf().size()) && !fill()) return false; b = buf().data() + pos; avail = buf().size() - pos; return true; }
    void advance(size_t n) { pos += n; }
    size_t offset() const { return 0; }
};

// libdeflate (whole-buffer inflate, 2-3x zlib's rate) when the system has the shared library; its four entry points are declared here
// because the image carries no header for it.  Absent: zlib.
struct Deflate {
    void *(*alloc)() = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*free_)(void *) = nullptr;
    unsigned (*crc)(unsigned, const void *, size_t) = nullptr;
    int (*gzip_ex)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;   // (one whole gzip member, CRC and size checked)
    Deflate() {
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        free_ = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        crc = (unsigned (*)(unsigned, const void *, size_t))dlsym(h, "libdeflate_crc32");
        gzip_ex = (int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *))dlsym(h, "libdeflate_gzip_decompress_ex");
        if (!alloc || !decompress || !free_ || !crc) alloc = nullptr;
    }
    bool ok() const { return alloc != nullptr; }
};
const Deflate &deflate_lib() {
    static Deflate d;
    static const bool told = [] { if (d.ok() && !getenv("KCT_NO_LIBDEFLATE")) pgz::crc_impl() = d.crc; return true; }();   // (parallel_inflate.h's checksums)
    (void)told;
    return d;
}
bool libdeflate_disabled() { return getenv("KCT_NO_LIBDEFLATE") != nullptr; }   // (the switch: tests and bench run zlib's inflate too)

struct Mapping {
    const unsigned char *p = nullptr;
    size_t size = 0;
    // (cheap by the time it runs: the pages were dropped by the parsers, segment by segment -- see parse_text)
    ~Mapping() { release(); }
    void release() { if (p) munmap((void *)p, size); p = nullptr; size = 0; }
};

// The file reader's device calls (uploads, the staging kernel) are made by ONE long-lived thread: a thread that has made HIP calls
// takes ~3.5 ms to exit (its runtime state is torn down), which a worker thread per call paid inside every call.  A second caller at
// the same time (another table, another thread) gets a thread of its own, as before.
struct DeviceThread {
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool busy = false, has_job = false, started = false;
    void loop() {
        for (;;) {
            std::function<void()> f;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return has_job; }); f = std::move(job); has_job = false; }
            f();
            { std::lock_guard<std::mutex> lk(mu); busy = false; }
            cv.notify_all();
        }
    }
    bool start(std::function<void()> f) {   // false: busy with another call (or no thread to be had)
        std::lock_guard<std::mutex> lk(mu);
        if (busy) return false;
        if (!started) { try { std::thread([this] { loop(); }).detach(); } catch (...) { return false; } started = true; }
        job = std::move(f); has_job = true; busy = true;
        cv.notify_all();
        return true;
    }
    void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !busy; }); }
};
DeviceThread &device_thread() {   // (never destroyed: its thread outlives main; a forked child starts with none)
    static DeviceThread *d = [] {
        DeviceThread *q = new DeviceThread;
        (void)pthread_atfork(nullptr, nullptr, [] { DeviceThread *c = &device_thread(); new (c) DeviceThread; });
        return q;
    }();
    return *d;
}

}  // namespace

extern "C" const char *kct_inflater_name(void) { return deflate_lib().ok() && !libdeflate_disabled() ? "libdeflate" : "zlib"; }

extern "C" kct_status kct_consume_file(kct_table *t, const char *path, int skip_bad, uint64_t *n_total, uint64_t *n_records,
                                       uint64_t *n_bases) {
    KCT_BORROW(t);
    KCT_DBG(t, "file: call begins\n");
    KCT_TRY(use_consume(t));
    if (!path || !n_total) { set_err("null argument"); return KCT_ERR_ARG; }
    if (!skip_bad) { set_err("kct_consume_file supports skip_bad_kmers=True only; use kct_consume_batch for error mode"); return KCT_ERR_ARG; }
    *n_total = 0;
    if (n_records) *n_records = 0;
    if (n_bases) *n_bases = 0;

    // plain or gzip?  Plain files of some size are mapped and parsed by several threads.
    Mapping map;
    bool gz = false;
    {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) { set_err("cannot open %s", path); return KCT_ERR_ARG; }
        struct stat sb;
        unsigned char magic[2] = {0, 0};
        const bool regular = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
        if (regular && sb.st_size >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) gz = true;
        if (regular && !gz && sb.st_size > 0) {
            void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                map.p = (const unsigned char *)m; map.size = (size_t)sb.st_size;
                (void)madvise(m, map.size, MADV_SEQUENTIAL);
            }
        }
        close(fd);
    }
    const size_t k = t->k;
    KCT_DBG(t, "file: opened%s\n", map.p ? " and mapped" : "");
    // gzip: BGZF or a plain stream?
    Mapping gzmap;
    std::vector<BgzfBlock> blocks;
    if (gz) {
        const int fd = open(path, O_RDONLY);
        struct stat sb;
        if (fd >= 0 && fstat(fd, &sb) == 0 && sb.st_size > 0) {
            void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { gzmap.p = (const unsigned char *)m; gzmap.size = (size_t)sb.st_size; }
        }
        if (fd >= 0) close(fd);
    }
    const bool bgzf = gzmap.p && bgzf_scan(gzmap.p, gzmap.size, blocks);
    // A single-member gzip file whose text (the trailer's size field) is small is inflated in ONE piece by libdeflate -- about twice
    // zlib's streaming rate -- and then parsed like a mapped plain file; larger ones, several members, or no libdeflate: the stream below.
    Mapping whole;   // (anonymous pages, unmapped when the call returns)
    const unsigned char *text_p = map.p;
    size_t text_size = map.size;
    const bool have_ld = deflate_lib().ok() && deflate_lib().gzip_ex && !getenv("KCT_NO_LIBDEFLATE");
    const bool try_parallel = gzmap.size >= ((size_t)4 << 20) && !getenv("KCT_NO_PARALLEL_GZIP");
    if (gz && !bgzf && gzmap.p && gzmap.size > 18 && (have_ld || try_parallel)) {
        unsigned isize;
        memcpy(&isize, gzmap.p + gzmap.size - 4, 4);
        size_t limit = (size_t)2 << 30;
        if (const char *e = getenv("KCT_GZIP_WHOLE_MAX")) limit = (size_t)atoll(e);
        if (isize && isize <= limit) {
            const size_t bytes = ((size_t)isize + 16 + 4095) & ~(size_t)4095;
            void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m != MAP_FAILED) {
                whole.p = (const unsigned char *)m; whole.size = bytes;
                (void)madvise(m, bytes, MADV_HUGEPAGE);   // (where transparent huge pages are on request: 512x fewer faults for the inflaters)
                // the pages are faulted in by four helper threads while the inflater runs (a fresh page costs about as much as inflating
                // it: left to the inflater, the faults were a sixth of the call)
                std::vector<std::thread> toucher;
                const size_t quarter = ((bytes / 4) + 4095) & ~(size_t)4095;
                for (size_t off = 0; off < bytes; off += quarter)
                    toucher.emplace_back([=] {
                        const size_t n = std::min(quarter, bytes - off);
                        if (madvise((char *)m + off, n, 23 /* MADV_POPULATE_WRITE */) != 0)
                            for (size_t i = 0; i < n; i += 4096) ((volatile char *)m)[off + i] = 0;
                    });
                // Several threads on the ONE member where it is large enough (parallel_inflate.h: entered at block boundaries found by
                // search, the unknown 32 KiB in front of each piece resolved afterwards, length and CRC-32 verified); declined or failed:
                // libdeflate inflates it in one piece, as before.
                bool ok = false;
                if (try_parallel) {
                    const unsigned hw = std::thread::hardware_concurrency();
                    unsigned nth = std::max(2u, std::min(32u, hw / 2));   // (64 threads: no faster inflating and an erratic last phase, tools/gz_diag.py)
                    if (const char *e = getenv("KCT_GZIP_THREADS")) nth = (unsigned)std::max(2, atoi(e));
                    try { ok = pgz::gunzip_parallel(gzmap.p, gzmap.size, (uint8_t *)m, isize, nth); }
                    catch (...) { ok = false; }   // (no thread, no memory: the ordinary inflater takes the member; nothing unwinds across the C ABI)
                    KCT_DBG(t, "file: parallel inflate of %zu -> %u bytes on %u threads: %s\n", gzmap.size, isize, nth, ok ? "ok" : "declined");
                }
                if (!ok && have_ld) {
                    const Deflate &ld = deflate_lib();
                    void *dec = ld.alloc();
                    size_t n_in = 0, n_out = 0;
                    ok = dec && ld.gzip_ex(dec, gzmap.p, gzmap.size, m, isize, &n_in, &n_out) == 0 && n_in == gzmap.size && n_out == isize;
                    if (dec) ld.free_(dec);
                }
                KCT_DBG(t, "file: inflated\n");
                for (auto &th : toucher) th.join();
                KCT_DBG(t, "file: output pages populated\n");
                if (ok) { text_p = whole.p; text_size = isize; }
                // (else: more members, a size field that wrapped, or a corrupt file -- the streaming reader finds out)
            }
        }
    }
    // stream bytes per chunk (16 / 8 MiB until round 6, when every chunk cost a wait: with 4 MiB the link has work 0.3 ms into the call and never idles)
    // -- unless the table is so small that it will not stage a chunk of that size behind the others but count it by itself (kct_entry.hip:
    // a piece of four window starts per slot or more): 16 MiB then, as before, for passes worth their launches
    const bool small_table = 4 * (size_t)t->cap <= ((size_t)4 << 20);
    size_t chunk_cap = (size_t)(small_table ? 16 : 4) << 20;
    if (const char *e = getenv("KCT_FILE_CHUNK")) chunk_cap = std::max<size_t>(1024, (size_t)atoll(e));  // tests shrink it to exercise record splitting
    size_t segment = (size_t)8 << 20;     // file bytes a parser thread takes at a time
    if (const char *e = getenv("KCT_FILE_SEGMENT")) segment = std::max<size_t>(64, (size_t)atoll(e));
    size_t slot_bytes = (size_t)(bgzf ? 1 : 4) << 20;   // inflated text per slot (BGZF: 2 MiB until round 6 -- 80 tasks over 32 threads on the C2 file; 1 MiB: +11 %)
    if (const char *e = getenv("KCT_FILE_SLOT")) slot_bytes = std::max<size_t>(65536, (size_t)atoll(e));  // tests shrink it: records across slots
    const unsigned hw = std::thread::hardware_concurrency();
    // BGZF tasks = runs of blocks whose text fits a slot, numbered in file order
    std::vector<size_t> task_first;
    if (bgzf) {
        size_t text = 0;
        for (size_t i = 0; i < blocks.size(); ++i) {
            if (i == 0 || text + blocks[i].isize > slot_bytes) { task_first.push_back(i); text = 0; }
            text += blocks[i].isize;
        }
        task_first.push_back(blocks.size());
    }
    const size_t ntasks = bgzf ? task_first.size() - 1 : 0;
    size_t nparsers = 1, nslot_threads = 0;
    // Round 6: a gzip file of 4 MiB or more that was not inflated in one piece goes through parallel_inflate.h's streaming form (below)
    const bool stream_parallel = gz && !bgzf && !text_p && gzmap.p && gzmap.size >= ((size_t)4 << 20) && !getenv("KCT_NO_PARALLEL_GZIP") &&
                                 pgz::gzip_header(gzmap.p, gzmap.size) != 0;
    if (text_p) nparsers = std::max<size_t>(1, std::min<size_t>({(size_t)8, hw ? hw : 1, (text_size + segment - 1) / segment}));
    if (stream_parallel) nparsers = std::max<size_t>(1, std::min<size_t>((size_t)8, hw ? hw : 1));
    if (bgzf) nslot_threads = std::max<size_t>(1, std::min<size_t>({(size_t)32, hw ? hw / 2 : 1, ntasks}));   // (16 until round 6: 5.8x10^9 k-mers/s on the C2 file against 9.1 with 32)
    if (const char *e = getenv("KCT_FILE_THREADS")) {
        const size_t want = std::max<size_t>(1, std::min<size_t>(64, (size_t)atoll(e)));
        if (text_p || stream_parallel) nparsers = want;
        if (bgzf) nslot_threads = want;
    }
    if (bgzf) nparsers = nslot_threads + 1;   // (+ the calling thread, which parses the fragments between the slots)

    // chunk buffers stay with the table: pinning memory costs more than parsing a small file
    if (t->h_file.size() < 2 * nparsers) t->h_file.resize(2 * nparsers);
    for (size_t i = 0; i < 2 * nparsers; ++i) KCT_TRY(t->h_file[i].reserve(chunk_cap + 64));
    // (the device side of a chunk: two halves that take turns -- chunk i + 1 is copied while chunk i is staged)
    const size_t half = (chunk_cap + 64 + 255) & ~(size_t)255;
    KCT_TRY(t->d_stream.reserve(2 * half));
    if (!t->copy_stream && hipStreamCreateWithFlags(&t->copy_stream, hipStreamNonBlocking) != hipSuccess) { set_err("hipStreamCreate failed"); return KCT_ERR_HIP; }

    ChunkQueue queue;
    std::deque<std::pair<FileChunk *, hipEvent_t>> pending;   // (the worker's: chunks whose copies are enqueued, oldest first)
    auto worker_body = [&] {  // uploads finished chunks and stages them behind one another on the device, one at a time
        (void)hipSetDevice(t->device);
        // Nothing is waited for per chunk (0.05 ms of round trips each, as much as a 2 MiB chunk's copy): a chunk's pinned buffer goes back to
        // its parser when the event behind its copy has passed -- looked at after the NEXT chunk's copy is under way -- and the chunks' good
        // windows add up in a device word read once at the end.  (A chunk the table will not stage right now -- the staged pieces are due
        // to be counted first -- takes the waiting form.)
        // The copies run on a stream of their own into the two halves of d_stream in turn: a half is copied into once the staging
        // kernel that read it last has run (staged_ev), and a chunk is staged once its copy has run (the event that also releases its buffer).
        std::vector<hipEvent_t> spare;
        spare.reserve(2 * nparsers + 8);   // (no more events than chunks exist: release_oldest's push_back never allocates)
        hipEvent_t staged_ev[2] = {nullptr, nullptr};
        bool staged_set[2] = {false, false};
        u64 total = 0, nchunk = 0;
        kct_status ws = KCT_OK;
        for (auto &e : staged_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { set_err("hipEventCreate failed"); ws = KCT_ERR_HIP; }
        auto release_oldest = [&] {
            auto pr = pending.front(); pending.pop_front();
            if (hipEventSynchronize(pr.second) != hipSuccess && ws == KCT_OK) { set_err("waiting for an upload failed"); ws = KCT_ERR_HIP; }
            spare.push_back(pr.second);
            { std::lock_guard<std::mutex> lk(queue.mu); pr.first->in_flight = false; }
            queue.cv.notify_all();
        };
        auto collect = [&] {   // the device word into `total` (everything enqueued so far has run), and zero again
            if (ws != KCT_OK) return;
            if (hipMemcpyAsync(t->h_counters, staged_good_word(t), 8, hipMemcpyDeviceToHost, t->stream) != hipSuccess || hipStreamSynchronize(t->stream) != hipSuccess ||
                hipMemsetAsync(staged_good_word(t), 0, 8, t->stream) != hipSuccess) { set_err("reading the staged windows' count failed"); ws = KCT_ERR_HIP; return; }
            total += t->h_counters[0];
        };
        if (hipMemsetAsync(staged_good_word(t), 0, 8, t->stream) != hipSuccess) { set_err("hipMemsetAsync failed"); ws = KCT_ERR_HIP; }
        for (;;) {
            FileChunk *c = nullptr;
            {
                std::unique_lock<std::mutex> lk(queue.mu);
                if (queue.q.empty() && !queue.done && !pending.empty()) {   // (before sleeping: a parser may be waiting for one of these)
                    lk.unlock();
                    while (!pending.empty()) release_oldest();
                    continue;
                }
                queue.cv.wait(lk, [&] { return !queue.q.empty() || queue.done; });
                if (queue.q.empty()) break;
                c = queue.q.front(); queue.q.pop_front();
            }
            const double t_take = now_ms();
            bool queued = false;
            static const bool dry = getenv("KCT_FILE_DRY") != nullptr;   // (measurement: the parsers alone -- nothing is uploaded or counted)
            if (ws == KCT_OK && !queue.failed() && !dry) {
                const size_t padded = (c->used + 15) & ~(size_t)15;
                memset((char *)c->host->p + c->used, '\n', padded + 16 - c->used);
                hipEvent_t ev = nullptr;
                if (!spare.empty()) { ev = spare.back(); spare.pop_back(); }
                else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_err("hipEventCreate failed"); ws = KCT_ERR_HIP; }
                const int hb = (int)(nchunk++ & 1);
                const unsigned char *d_half = (const unsigned char *)t->d_stream.p + hb * half;
                if (ws == KCT_OK && ((staged_set[hb] && hipStreamWaitEvent(t->copy_stream, staged_ev[hb], 0) != hipSuccess) ||
                                     hipMemcpyAsync((void *)d_half, c->host->p, padded + 16, hipMemcpyHostToDevice, t->copy_stream) != hipSuccess ||
                                     hipEventRecord(ev, t->copy_stream) != hipSuccess ||
                                     hipStreamWaitEvent(t->stream, ev, 0) != hipSuccess)) { set_err("H2D copy failed"); ws = KCT_ERR_HIP; }
                if (ev) { pending.emplace_back(c, ev); queued = true; }
                // (staged behind the earlier chunks and counted with them in passes of the table's own choosing: kct_entry.hip)
                bool staged = false;
                if (ws == KCT_OK) ws = stage_piece_async(t, d_half, c->used, &staged);
                if (ws == KCT_OK && !staged) {
                    collect();
                    u64 n = 0;
                    if (ws == KCT_OK) ws = consume_device_staged(t, d_half, c->used, &n);
                    total += n;
                    if (ws == KCT_OK && hipMemsetAsync(staged_good_word(t), 0, 8, t->stream) != hipSuccess) { set_err("hipMemsetAsync failed"); ws = KCT_ERR_HIP; }
                }
                if (ws == KCT_OK) { if (hipEventRecord(staged_ev[hb], t->stream) != hipSuccess) { set_err("hipEventRecord failed"); ws = KCT_ERR_HIP; } else staged_set[hb] = true; }
                if (ws != KCT_OK) queue.fail(ws, g_err);
            }
            if (!queued) {
                { std::lock_guard<std::mutex> lk(queue.mu); c->in_flight = false; }
                queue.cv.notify_all();
            }
            const double t_enq = now_ms();
            while (pending.size() > 2) release_oldest();   // (two copies stay enqueued: the link always has the next one)
            KCT_DBG(t, "file: worker took a chunk of %zu bytes at %.3f: enqueued in %.0f us, then waited %.0f us for an older copy\n", c->used, t_take, (t_enq - t_take) * 1e3,
                    (now_ms() - t_enq) * 1e3);
        }
        while (!pending.empty()) release_oldest();
        collect();
        if (ws != KCT_OK) queue.fail(ws, g_err);
        for (hipEvent_t e : spare) (void)hipEventDestroy(e);
        for (hipEvent_t e : staged_ev) if (e) (void)hipEventDestroy(e);
        { std::lock_guard<std::mutex> lk(queue.mu); queue.counted += total; }
        KCT_DBG(t, "file: worker is through\n");
    };
    auto worker_fn = [&] {   // (nothing may leave a thread by exception: out of memory in the worker's few containers fails the call instead)
        try { worker_body(); }
        catch (...) {
            queue.fail(KCT_ERR_HIP, "out of memory in the file reader's worker");
            (void)hipStreamSynchronize(t->copy_stream);
            (void)hipStreamSynchronize(t->stream);
            for (;;) {   // the parsers' chunks come back unread until they have all stopped
                std::unique_lock<std::mutex> lk(queue.mu);
                for (auto &pr : pending) pr.first->in_flight = false;   // (their copies have ended: the streams were waited for)
                pending.clear();
                for (FileChunk *c : queue.q) c->in_flight = false;
                queue.q.clear();
                queue.cv.notify_all();
                if (queue.done) break;
                queue.cv.wait(lk, [&] { return !queue.q.empty() || queue.done; });
            }
        }
    };
    std::thread own_worker;   // (only when the long-lived device thread is taken by another call)
    const bool shared_worker = device_thread().start(worker_fn);
    if (!shared_worker) own_worker = std::thread(worker_fn);
    struct WorkerEnd {   // whatever way this scope is left, the worker has ended before `queue` and the rest go away
        ChunkQueue &q; std::thread &own; bool shared, ended = false;
        void end() {
            if (ended) return;
            ended = true;
            { std::lock_guard<std::mutex> lk(q.mu); q.done = true; }
            q.cv.notify_all();
            if (shared) device_thread().wait(); else if (own.joinable()) own.join();
        }
        ~WorkerEnd() { end(); }
    } worker_end{queue, own_worker, shared_worker};

    u64 records = 0, bases = 0;
    kct_status st = KCT_OK;
    // [p, p + size) holds whole records (it begins at a record start): parsed by nparsers threads that take `segment`-sized stretches (record
    // starts found without context) -- a mapped file, the one-piece inflate of a .gz, or one window of the streaming inflater's text
    // (MADV_POPULATE_READ on a mapped file's segments before they are parsed: measured, no gain -- the page faults are not what holds the parsers)
    auto parse_text = [&](const unsigned char *p, size_t size, int fmt) {
        const size_t nseg = (size + segment - 1) / segment;
        std::atomic<size_t> next_seg{0};
        std::mutex tally_mu;
        auto parser = [&](size_t id) {
            ChunkWriter w(&queue, &t->h_file[2 * id], &t->h_file[2 * id + 1], chunk_cap, k);
            for (;;) {
                const size_t sg = next_seg.fetch_add(1);
                if (sg >= nseg || queue.failed()) break;
                const size_t lo = find_record_start(p, size, sg * segment, fmt);
                const size_t hi = sg + 1 == nseg ? size : find_record_start(p, size, (sg + 1) * segment, fmt);
                if (lo >= hi) continue;  // no record starts in this segment
                MemSource src{p, p + lo, p + size};
                if (!parse_records(src, w, fmt, hi, queue, path)) { queue.fail(KCT_ERR_ARG, g_err); break; }
                // A mapped file's pages leave the page table here, segment by segment and thread by thread (MADV_DONTNEED takes the address
                // space's lock for reading): unmapping the 160 MB of the C2 file in one piece took 2.2 ms inside the call -- or, handed to a
                // thread, held the lock against the NEXT call's mmap for 5.9 ms.  [lo, hi) is read by this thread only.
                // (the anonymous text of a .gz inflated in one piece likewise: its pages are freed here; file_gz's 627 MB sample -8 %)
                static const bool no_zap = getenv("KCT_NO_ZAP") != nullptr;   // (measurement)
                if (!no_zap && (p == map.p || p == whole.p) && hi > lo) {
                    const uintptr_t a0 = ((uintptr_t)p + lo + 4095) & ~(uintptr_t)4095, a1 = ((uintptr_t)p + hi) & ~(uintptr_t)4095;
                    if (a1 > a0) (void)madvise((void *)a0, a1 - a0, MADV_DONTNEED);
                }
            }
            w.finish();
            std::lock_guard<std::mutex> lk(tally_mu);
            records += w.records; bases += w.bases;
        };
        std::vector<std::thread> pool;
        for (size_t i = 1; i < nparsers; ++i) pool.emplace_back(parser, i);
        parser(0);
        for (auto &th : pool) th.join();
    };
    if (text_p) {
        size_t first = 0;
        while (first < text_size && (text_p[first] == '\n' || text_p[first] == '\r' || text_p[first] == ' ' || text_p[first] == '\t')) ++first;
        const int fmt = first < text_size ? text_p[first] : '>';
        if (first < text_size && fmt != '>' && fmt != '@') { set_err("%s: neither FASTA nor FASTQ (starts with 0x%02x)", path, fmt); st = KCT_ERR_ARG; }
        if (st == KCT_OK && first < text_size) {
            parse_text(text_p, text_size, fmt);
            KCT_DBG(t, "file: parsers done\n");
        }
    } else if (bgzf) {
        // the file's first byte of text says FASTA or FASTQ (the slot threads need to know before they meet it)
        int fmt = -1;
        {
            std::vector<unsigned char> first(65536);
            for (size_t i = 0; i < blocks.size() && fmt < 0 && st == KCT_OK; ++i) {
                const BgzfBlock &b = blocks[i];
                const unsigned char *hdr = gzmap.p + b.off;
                const size_t data = 12 + (hdr[10] | ((size_t)hdr[11] << 8));
                z_stream zs;
                memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, -15) != Z_OK) { set_err("inflateInit2 failed"); st = KCT_ERR_ARG; break; }
                zs.next_in = const_cast<unsigned char *>(hdr + data); zs.avail_in = (unsigned)(b.csize - data - 8);
                zs.next_out = first.data(); zs.avail_out = (unsigned)first.size();
                const int rc = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (rc != Z_STREAM_END || zs.total_out != b.isize) { set_err("%s: corrupt BGZF block", path); st = KCT_ERR_ARG; break; }
                for (size_t j = 0; j < b.isize && fmt < 0; ++j)
                    if (first[j] != '\n' && first[j] != '\r' && first[j] != ' ' && first[j] != '\t') fmt = first[j];
            }
            if (st == KCT_OK && fmt >= 0 && fmt != '>' && fmt != '@') { set_err("%s: neither FASTA nor FASTQ (starts with 0x%02x)", path, fmt); st = KCT_ERR_ARG; }
        }
        if (st == KCT_OK && fmt >= 0) {
            if (t->file_text.size() < nparsers) t->file_text.resize(nparsers);   // (indexed like h_file: by the thread's id)
            for (size_t i = 0; i < nparsers; ++i) if (t->file_text[i].size() < slot_bytes + 65536 + 16) t->file_text[i].resize(slot_bytes + 65536 + 16);
            Fragments frags(2 * nslot_threads + 2, ntasks);
            std::atomic<size_t> next_task{0};
            std::mutex tally_mu;
            auto slot_thread = [&](size_t id) {
                ChunkWriter w(&queue, &t->h_file[2 * id], &t->h_file[2 * id + 1], chunk_cap, k);
                // (the slot's text buffer stays with the table: thirty-two threads each mapping, zero-filling and unmapping 2 MiB per call
                // met in the address space's lock)
                std::vector<unsigned char> &text = t->file_text[id];
                const Deflate &ld = deflate_lib();
                void *dec = ld.ok() && !getenv("KCT_NO_LIBDEFLATE") ? ld.alloc() : nullptr;   // (the switch: tests run zlib's inflate too)
                z_stream zs;
                memset(&zs, 0, sizeof zs);
                if (!dec && inflateInit2(&zs, -15) != Z_OK) { frags.fail("inflateInit2 failed"); return; }
                for (;;) {
                    const size_t task = next_task.fetch_add(1);
                    if (task >= ntasks || !frags.may_start(task)) break;
                    if (queue.failed()) { frags.fail("counting failed"); break; }
                    size_t used = 0;
                    bool ok = true;
                    for (size_t i = task_first[task]; ok && i < task_first[task + 1]; ++i) {
                        const BgzfBlock &b = blocks[i];
                        const unsigned char *hdr = gzmap.p + b.off;
                        const size_t data = 12 + (hdr[10] | ((size_t)hdr[11] << 8));
                        unsigned crc, got;
                        memcpy(&crc, hdr + b.csize - 8, 4);
                        if (dec) {
                            size_t n_out = 0;
                            ok = ld.decompress(dec, hdr + data, b.csize - data - 8, text.data() + used, b.isize, &n_out) == 0 && n_out == b.isize;
                            got = ok ? ld.crc(0, text.data() + used, b.isize) : 0;
                        } else {
                            inflateReset(&zs);
                            zs.next_in = const_cast<unsigned char *>(hdr + data); zs.avail_in = (unsigned)(b.csize - data - 8);
                            zs.next_out = text.data() + used; zs.avail_out = (unsigned)b.isize;
                            ok = inflate(&zs, Z_FINISH) == Z_STREAM_END && zs.total_out == b.isize;
                            got = ok ? (unsigned)crc32(0L, text.data() + used, b.isize) : 0;
                        }
                        ok = ok && got == crc;
                        used += b.isize;
                    }
                    if (!ok) { frags.fail("corrupt BGZF block"); break; }
                    {   // (the task's compressed bytes leave the page table here, thread by thread: see parse_text)
                        const BgzfBlock &b0 = blocks[task_first[task]], &b1 = blocks[task_first[task + 1] - 1];
                        const uintptr_t a0 = ((uintptr_t)gzmap.p + b0.off + 4095) & ~(uintptr_t)4095, a1 = ((uintptr_t)gzmap.p + b1.off + b1.csize) & ~(uintptr_t)4095;
                        if (a1 > a0) (void)madvise((void *)a0, a1 - a0, MADV_DONTNEED);
                    }
                    const unsigned char *tx = text.data();
                    size_t lo = 0;
                    if (task == 0) { while (lo < used && (tx[lo] == '\n' || tx[lo] == '\r' || tx[lo] == ' ' || tx[lo] == '\t')) ++lo; }  // the file's first record
                    else lo = find_record_start(tx, used, 1, fmt);
                    if (lo >= used) { frags.publish(task, tx, used, nullptr, 0); continue; }   // no record starts in this slot
                    const size_t hi = find_last_record_start(tx, used, lo, fmt);
                    if (hi > lo) {
                        MemSource src{tx, tx + lo, tx + hi};
                        if (!parse_records(src, w, fmt, hi, queue, path)) { queue.fail(KCT_ERR_ARG, g_err); frags.fail(g_err); break; }
                    }
                    frags.publish(task, tx, lo, tx + hi, used - hi);
                }
                if (dec) ld.free_(dec); else inflateEnd(&zs);
                w.finish();
                std::lock_guard<std::mutex> lk(tally_mu);
                records += w.records; bases += w.bases;
            };
            std::vector<std::thread> pool;
            for (size_t i = 1; i <= nslot_threads; ++i) pool.emplace_back(slot_thread, i);
            {
                FragmentSource fsrc{&frags};
                ChunkWriter w0(&queue, &t->h_file[0], &t->h_file[1], chunk_cap, k);
                if (!parse_records(fsrc, w0, fmt, ~(size_t)0, queue, path)) st = KCT_ERR_ARG;
                w0.finish();
                std::lock_guard<std::mutex> lk(tally_mu);
                records += w0.records; bases += w0.bases;
            }
            bool slot_failed;
            std::string slot_msg;
            { std::lock_guard<std::mutex> lk(frags.mu); slot_failed = frags.failed; slot_msg = frags.msg; }
            frags.fail("the parser is done");   // (lets slot threads go that still wait for their turn: the parser stopped early)
            for (auto &th : pool) th.join();
            KCT_DBG(t, "file: parsers done\n");
            if (st == KCT_OK && slot_failed) { set_err("%s: %s", path, slot_msg.c_str()); st = KCT_ERR_ARG; }
        }
    } else if (stream_parallel) {
        // A plain gzip stream too large (or of too many members) for the one-piece route: 32 MiB of compressed bytes at a time are inflated by
        // several threads (entered at searched block boundaries; every member's CRC-32 and length checked at its end -- a mismatch fails the
        // call there, as gzread's would), the next window while this one's text is parsed by nparsers threads like a mapped file.  What
        // follows a window's last record start -- an unfinished record -- is carried in front of the next window's text.
        KCT_DBG(t, "file: gzip stream through the parallel inflater (windows of 32 MiB), %zu parser threads\n", nparsers);
        std::string fail_msg;
        try {
            unsigned nth = std::max(2u, std::min(32u, hw / 2));
            if (const char *e = getenv("KCT_GZIP_THREADS")) nth = (unsigned)std::max(2, atoi(e));
            size_t span = (size_t)32 << 20;
            if (const char *e = getenv("KCT_GZIP_WINDOW")) span = std::max<size_t>(65536, (size_t)atoll(e));   // (tests: many windows of a small file)
            size_t off = 0;
            pgz::MemberStream ms;
            bool in_member = false;
            // the next window's text (members follow one another; what is no gzip header behind a member's trailer is ignored, as zlib does)
            // (the window buffers -- some 700 MB of mappings by the end -- are unmapped by a thread of their own after the call: 45 ms otherwise)
            struct Buffers { pgz::WindowScratch scratch; pgz::TextBuf text, nxt; };
            std::shared_ptr<Buffers> bufs = std::make_shared<Buffers>();
            struct Release { std::shared_ptr<Buffers> &b; ~Release() { try { std::thread([q = std::move(b)]() mutable { q.reset(); }).detach(); } catch (...) {} } } release{bufs};
            pgz::WindowScratch &scratch = bufs->scratch;
            auto next_text = [&](pgz::TextBuf &text, std::string &msg) -> int {   // 1 = text, 0 = the end, -1 = failed
                for (;;) {
                    if (!in_member) {
                        const size_t h = off < gzmap.size ? pgz::gzip_header(gzmap.p + off, gzmap.size - off) : 0;
                        if (!h) return 0;
                        ms = pgz::MemberStream();
                        ms.def = gzmap.p + off + h; ms.def_size = gzmap.size - off - h;
                        off += h;
                        in_member = true;
                    }
                    if (!pgz::inflate_window(ms, span, nth, text, scratch)) { msg = "corrupt gzip stream"; return -1; }
                    if (ms.done) {
                        const size_t end = (size_t)((ms.bit + 7) / 8);
                        unsigned crc = 0, isz = 0;
                        if (end + 8 > ms.def_size) { msg = "truncated gzip stream"; return -1; }
                        memcpy(&crc, ms.def + end, 4); memcpy(&isz, ms.def + end + 4, 4);
                        if (crc != ms.crc || isz != (unsigned)ms.total) { msg = "corrupt gzip stream (CRC-32 / length of a member)"; return -1; }
                        off += end + 8;
                        in_member = false;
                    }
                    if (!text.empty()) return 1;
                }
            };
            // Two text buffers take turns (their pages stay); what a window leaves unparsed -- the bytes from its last record start on -- is
            // copied into the free `lead` in front of the next window's text, or, where a record is longer than that, joined with it aside.
            pgz::TextBuf &text = bufs->text, &nxt = bufs->nxt;
            text.lead = nxt.lead = (size_t)1 << 20;
            if (const char *e = getenv("KCT_GZIP_LEAD")) text.lead = nxt.lead = (size_t)atoll(e);   // (tests: the joined route)
            std::vector<unsigned char> carry, joined;
            int fmt = -1;
            int have = next_text(text, fail_msg);
            while (have == 1 && !queue.failed()) {
                int have_next = 0;
                std::string next_msg;
                std::thread ahead([&] { try { have_next = next_text(nxt, next_msg); } catch (...) { next_msg = "out of memory in the parallel inflater"; have_next = -1; } });
                struct Joiner { std::thread &th; ~Joiner() { if (th.joinable()) th.join(); } } joiner{ahead};   // (an exception below must not leave it running)
                const unsigned char *p = text.data();
                size_t size = text.size();
                if (!carry.empty()) {
                    if (carry.size() <= text.lead) { p -= carry.size(); memcpy(text.data() - carry.size(), carry.data(), carry.size()); }
                    else { joined.assign(carry.begin(), carry.end()); joined.insert(joined.end(), text.data(), text.data() + text.size()); p = joined.data(); }
                    size += carry.size();
                }
                if (fmt < 0) {
                    while (size && (*p == '\n' || *p == '\r' || *p == ' ' || *p == '\t')) { ++p; --size; }
                    if (size) {
                        fmt = *p;
                        if (fmt != '>' && fmt != '@') { set_err("%s: neither FASTA nor FASTQ (starts with 0x%02x)", path, fmt); st = KCT_ERR_ARG; }
                    }
                }
                if (st == KCT_OK && fmt >= 0) {
                    const size_t hi = find_last_record_start(p, size, 0, fmt);
                    if (hi > 0) {
                        KCT_DBG(t, "file: window of %zu text bytes to the parsers\n", hi);
                        parse_text(p, hi, fmt);
                        KCT_DBG(t, "file: window parsed\n");
                    }
                    std::vector<unsigned char> rest(p + hi, p + size);
                    carry.swap(rest);
                } else carry.clear();   // (white space only so far)
                ahead.join();
                KCT_DBG(t, "file: next window inflated\n");
                if (st != KCT_OK) { have = -2; break; }
                std::swap(text.p, nxt.p); std::swap(text.cap, nxt.cap); std::swap(text.n, nxt.n);
                have = have_next;
                fail_msg = next_msg;
            }
            if (have == 0 && st == KCT_OK && fmt >= 0 && !carry.empty() && !queue.failed()) parse_text(carry.data(), carry.size(), fmt);   // the last record(s)
            if (have == -1 && st == KCT_OK) { set_err("%s: %s", path, fail_msg.c_str()); st = KCT_ERR_ARG; }
        } catch (...) { if (st == KCT_OK) { set_err("%s: out of memory in the parallel inflater", path); st = KCT_ERR_ARG; } }
        KCT_DBG(t, "file: parsers done\n");
    } else {
        // (said ONCE per process, loudly: without libdeflate.so.0 a gzip file is inflated by one zlib thread at ~0.4 GB/s of text -- below
        // what the CPU reference path reads -- and nothing downstream can make up for it; bgzip the file, or install libdeflate)
        static std::atomic<bool> said{false};
        if (!(deflate_lib().ok() && !libdeflate_disabled()) && gzmap.size > (64u << 20) && !said.exchange(true))
            fprintf(stderr, "kct_consume_file: libdeflate.so.0 not found -- %s is inflated by zlib on one thread (~0.4 GB/s of text); "
                            "BGZF input (bgzip) is inflated on many threads, libdeflate doubles a single stream's rate\n", path);
        // a small plain gzip stream: one zlib thread -> a ring of text slots -> this thread's parser (see the top of the file)
        TextRing ring(4, slot_bytes);
        std::thread inflater([&] {
            gzFile f = gzopen(path, "rb");
            if (!f) { ring.fail("cannot open the gzip stream"); return; }
            gzbuffer(f, 1 << 20);
            u64 seq = 0;
            for (;;) {
                TextRing::Slot *sl = ring.acquire(seq);
                if (!sl) break;
                const int n = gzread(f, sl->buf.data(), (unsigned)sl->buf.size());
                if (n < 0) { ring.fail("gzread failed (corrupt gzip stream)"); break; }
                if (n == 0) { ring.finish(seq); break; }
                ring.publish(sl, (size_t)n);
                ++seq;
            }
            gzclose(f);
        });
        {
            RingSource src{&ring};
            int c;
            while ((c = src.peek()) >= 0 && (c == '\n' || c == '\r' || c == ' ' || c == '\t')) src.advance(1);
            if (c >= 0 && c != '>' && c != '@') { set_err("%s: neither FASTA nor FASTQ (starts with 0x%02x)", path, c); st = KCT_ERR_ARG; }
            if (st == KCT_OK && c >= 0) {
                ChunkWriter w(&queue, &t->h_file[0], &t->h_file[1], chunk_cap, k);
                if (!parse_records(src, w, c, ~(size_t)0, queue, path)) st = KCT_ERR_ARG;
                w.finish();
                records = w.records; bases = w.bases;
            }
            if (st == KCT_OK && ring.failed) { set_err("%s: %s", path, ring.msg.c_str()); st = KCT_ERR_ARG; }
        }
        ring.fail("the parser is done");   // (lets the inflater go if it still waits for a slot: the parser stopped early, or the input is read)
        inflater.join();
    }
    worker_end.end();
    KCT_DBG(t, "file: chunks uploaded and counted (or staged)\n");
    whole.release();
    KCT_DBG(t, "file: the inflated text unmapped\n");
    gzmap.release();
    map.release();
    KCT_DBG(t, "file: the file unmapped\n");
    if (st == KCT_OK && queue.status != KCT_OK) { st = queue.status; set_err("%s", queue.msg.c_str()); }
    if (st != KCT_OK) return st;
    t->consumed += bases;
    *n_total = queue.counted;
    if (n_records) *n_records = records;
    if (n_bases) *n_bases = bases;
    return KCT_OK;
}
