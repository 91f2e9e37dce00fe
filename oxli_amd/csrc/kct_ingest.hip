// kct_ingest.hip -- FASTA / FASTQ file ingestion (kct_consume_file): the caller side of the path.
#include "kct_internal.h"

using namespace kcth;

// ---- FASTA / FASTQ ingestion: the caller side of the path (README.md:89-99) ---------------------------
// The reference delegates parsing to screed and calls consume() once per record.  Here a host
// parser turns the file (plain or gzip) into record-stream chunks in pinned memory while a worker
// thread uploads and counts the previous chunk, so parsing and device work overlap.  A record longer
// than a chunk is cut with a (k-1)-base overlap, which keeps every window counted exactly once.
namespace {

struct FileChunk {
    PinnedBuf host;
    DevBuf dev;
    size_t used = 0;
};

struct RecordParser {
    gzFile f = nullptr;
    std::vector<unsigned char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    int fmt = 0;  // '>' FASTA, '@' FASTQ, 0 unknown yet
    bool fill() {
        if (eof) return false;
        int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n <= 0) { eof = true; return false; }
        pos = 0; end = (size_t)n;
        return true;
    }
    int peek() { if (pos >= end && !fill()) return -1; return buf[pos]; }
    int get() { int c = peek(); if (c >= 0) ++pos; return c; }
    void skip_line() { int c; while ((c = get()) >= 0 && c != '\n') {} }
};

}  // namespace

extern "C" kct_status kct_consume_file(kct_table *t, const char *path, int skip_bad, uint64_t *n_total, uint64_t *n_records,
                                       uint64_t *n_bases) {
    KCT_TRY(use(t));
    if (!path || !n_total) { set_err("null argument"); return KCT_ERR_ARG; }
    if (!skip_bad) { set_err("kct_consume_file supports skip_bad_kmers=True only; use kct_consume_batch for error mode"); return KCT_ERR_ARG; }
    *n_total = 0;
    if (n_records) *n_records = 0;
    if (n_bases) *n_bases = 0;
    RecordParser ps;
    ps.f = gzopen(path, "rb");
    if (!ps.f) { set_err("cannot open %s", path); return KCT_ERR_ARG; }
    gzbuffer(ps.f, 1 << 20);
    ps.buf.resize(1 << 22);
    const size_t k = t->k;
    size_t chunk_cap = (size_t)64 << 20;  // stream bytes per chunk
    if (const char *e = getenv("KCT_FILE_CHUNK")) chunk_cap = std::max<size_t>(1024, (size_t)atoll(e));  // tests shrink it to exercise record splitting
    FileChunk chunks[2];
    kct_status st = KCT_OK;
    for (auto &c : chunks) {
        if (st == KCT_OK) st = c.host.reserve(chunk_cap + 64);
        if (st == KCT_OK) st = c.dev.reserve(chunk_cap + 64);
    }
    // worker: uploads and counts chunk `job` while the parser fills the other one
    std::mutex mu;
    std::condition_variable cv;
    int job = -1;             // chunk index handed to the worker, -1 = none
    bool done = false, busy = false;
    kct_status worker_status = KCT_OK;
    std::string worker_msg;
    u64 counted = 0;
    std::thread worker([&] {
        (void)hipSetDevice(t->device);
        for (;;) {
            int j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return job >= 0 || done; });
                if (job < 0 && done) return;
                j = job; job = -1; busy = true;
            }
            FileChunk &c = chunks[j];
            kct_status ws = KCT_OK;
            const size_t padded = (c.used + 15) & ~(size_t)15;
            memset((char *)c.host.p + c.used, '\n', padded + 16 - c.used);
            if (hipMemcpyAsync(c.dev.p, c.host.p, padded + 16, hipMemcpyHostToDevice, t->stream) != hipSuccess) { set_err("H2D copy failed"); ws = KCT_ERR_HIP; }
            u64 n = 0;
            if (ws == KCT_OK) ws = consume_stream(t, (const unsigned char *)c.dev.p, c.used, &n);
            {
                std::lock_guard<std::mutex> lk(mu);
                counted += n;
                if (ws != KCT_OK && worker_status == KCT_OK) { worker_status = ws; worker_msg = g_err; }
                busy = false;
            }
            cv.notify_all();
        }
    });
    auto submit = [&](int j) {  // hand chunk j to the worker once it is idle
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return job < 0 && !busy; });
        job = j;
        lk.unlock();
        cv.notify_all();
    };
    int cur = 0;
    u64 records = 0, bases = 0;
    unsigned char *out = (unsigned char *)chunks[cur].host.p;
    size_t used = 0, rec_len = 0;  // rec_len = bases of the current record already emitted into this chunk run
    auto flush = [&](bool mid_record) {
        // keep the last k-1 bases of an unfinished record: they open the next chunk
        unsigned char tail[256];
        size_t ntail = 0;
        if (mid_record) { ntail = std::min(rec_len, k - 1); memcpy(tail, out + used - ntail, ntail); }
        chunks[cur].used = used;
        submit(cur);
        cur ^= 1;
        // submit() returned once the worker was idle, i.e. the other buffer's chunk is finished: it is free
        out = (unsigned char *)chunks[cur].host.p;
        memcpy(out, tail, ntail);
        used = ntail;
        rec_len = ntail;
    };
    auto emit = [&](const unsigned char *p, size_t n) {  // append sequence bytes of the current record
        while (n) {
            if (used + 1 >= chunk_cap) flush(true);
            const size_t take = std::min(n, chunk_cap - 1 - used);
            memcpy(out + used, p, take);
            used += take; rec_len += take; p += take; n -= take; bases += take;
        }
    };
    auto end_record = [&] {
        if (used + 1 >= chunk_cap) flush(true);
        out[used++] = '\n';
        rec_len = 0;
        ++records;
    };
    // copies the rest of the current line (without CR/LF) into the record; returns its length
    auto emit_line = [&]() -> size_t {
        size_t total = 0;
        for (;;) {
            if (ps.pos >= ps.end && !ps.fill()) break;
            const unsigned char *b = ps.buf.data() + ps.pos;
            const size_t avail = ps.end - ps.pos;
            const unsigned char *nl = (const unsigned char *)memchr(b, '\n', avail);
            size_t n = nl ? (size_t)(nl - b) : avail;
            size_t m = n;
            if (m && b[m - 1] == '\r') --m;
            emit(b, m); total += m;
            ps.pos += n + (nl ? 1 : 0);
            if (nl) break;
        }
        return total;
    };
    if (st == KCT_OK) {
        int c;
        while ((c = ps.peek()) >= 0) {
            if (c == '\n' || c == '\r' || c == ' ' || c == '\t') { ps.get(); continue; }
            if (ps.fmt == 0) {
                if (c != '>' && c != '@') { set_err("%s: neither FASTA nor FASTQ (starts with 0x%02x)", path, c); st = KCT_ERR_ARG; break; }
                ps.fmt = c;
            }
            if (c != ps.fmt) { set_err("%s: malformed record header near record %llu", path, (unsigned long long)records); st = KCT_ERR_ARG; break; }
            ps.skip_line();  // header
            size_t seq_len = 0;
            if (ps.fmt == '>') {
                while ((c = ps.peek()) >= 0 && c != '>') seq_len += emit_line();
            } else {
                while ((c = ps.peek()) >= 0 && c != '+') seq_len += emit_line();
                ps.skip_line();  // '+' line
                size_t q = 0;    // quality: as many characters as the sequence had
                while (q < seq_len && ps.peek() >= 0) {
                    int d = ps.get();
                    if (d != '\n' && d != '\r') ++q;
                }
                ps.skip_line();
            }
            end_record();
        }
        if (st == KCT_OK && used) { chunks[cur].used = used; submit(cur); }
    }
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return job < 0 && !busy; });
        done = true;
    }
    cv.notify_all();
    worker.join();
    gzclose(ps.f);
    for (auto &c : chunks) { c.host.release(); c.dev.release(); }
    if (st == KCT_OK && worker_status != KCT_OK) { st = worker_status; set_err("%s", worker_msg.c_str()); }
    if (st != KCT_OK) return st;
    t->consumed += bases;
    *n_total = counted;
    if (n_records) *n_records = records;
    if (n_bases) *n_bases = bases;
    return KCT_OK;
}

