// kct_store.hip -- save / load in the reference's wire format (lib.rs:269-322): the serde_json image of the
// struct, {"counts":{"<hash>":count,...},"ksize":..,"version":..,"consumed":..,"store_kmers":..,"hash_to_kmer":..},
// gzip level 1 (load also accepts plain JSON, as niffler does).  The counts object is the part that scales with the
// table (10^7 - 10^9 entries), so it is produced and parsed here; the few scalar members travel as text to and from
// the caller, which owns them (version string, store_kmers map).
//   save: device dump sorted by hash -> decimal text -> deflate, both done by several threads on consecutive pieces.
//         Every piece is a raw deflate stream that ends on a byte boundary (Z_SYNC_FLUSH) and does not look back
//         into the previous piece, so the pieces concatenate into ONE gzip member that any gzip reader accepts.
//   load: inflate (zlib), one pass over the text with a decimal parser, pairs merged into a fresh device table.
#include "kct_internal.h"

using namespace kcth;

namespace {

char *put_u64(char *p, u64 v) {  // decimal, no terminator
    char tmp[20];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

struct Piece {
    std::vector<unsigned char> out;
    uLong crc = 0;
    size_t text_len = 0;
    int err = Z_OK;
};

// text -> raw deflate ending with a sync flush (byte aligned, not final)
void deflate_piece(z_stream &zs, const std::string &text, Piece &pc) {
    pc.crc = crc32(crc32(0L, Z_NULL, 0), (const Bytef *)text.data(), (uInt)text.size());
    pc.text_len = text.size();
    pc.out.resize(deflateBound(&zs, (uLong)text.size()) + 16);
    deflateReset(&zs);
    zs.next_in = (Bytef *)text.data(); zs.avail_in = (uInt)text.size();
    zs.next_out = pc.out.data(); zs.avail_out = (uInt)pc.out.size();
    const int rc = deflate(&zs, Z_SYNC_FLUSH);
    pc.err = (rc == Z_OK && zs.avail_in == 0) ? Z_OK : (rc == Z_OK ? Z_BUF_ERROR : rc);
    pc.out.resize(pc.out.size() - zs.avail_out);
}

bool write_all(FILE *f, const void *p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }

// ---- a minimal JSON reader for the load side -------------------------------------------------------------------
struct Cursor {
    const char *p, *end;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    bool eat(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
};

// p at an opening quote: returns the raw string body [b, e) (escapes left alone), p after the closing quote
bool read_string(Cursor &c, const char *&b, const char *&e) {
    c.ws();
    if (c.p >= c.end || *c.p != '"') return false;
    b = ++c.p;
    while (c.p < c.end && *c.p != '"') { if (*c.p == '\\') ++c.p; ++c.p; }
    if (c.p >= c.end) return false;
    e = c.p++;
    return true;
}

bool read_u64(const char *b, const char *e, u64 &v) {
    if (b == e) return false;
    v = 0;
    for (; b < e; ++b) {
        if (*b < '0' || *b > '9') return false;
        const u64 d = (u64)(*b - '0');
        if (v > (~0ULL - d) / 10) return false;
        v = v * 10 + d;
    }
    return true;
}

// skips one JSON value of any kind
bool skip_value(Cursor &c) {
    c.ws();
    if (c.p >= c.end) return false;
    if (*c.p == '"') { const char *b, *e; return read_string(c, b, e); }
    if (*c.p == '{' || *c.p == '[') {
        int depth = 0;
        while (c.p < c.end) {
            const char ch = *c.p;
            if (ch == '"') { const char *b, *e; if (!read_string(c, b, e)) return false; continue; }
            if (ch == '{' || ch == '[') ++depth;
            if (ch == '}' || ch == ']') { --depth; if (depth == 0) { ++c.p; return true; } }
            ++c.p;
        }
        return false;
    }
    while (c.p < c.end && *c.p != ',' && *c.p != '}' && *c.p != ']' && *c.p != ' ' && *c.p != '\n' && *c.p != '\r' && *c.p != '\t') ++c.p;
    return true;
}

thread_local std::string g_rest;  // the members of the last loaded file other than counts, as one JSON object

}  // namespace

extern "C" {

const char *kct_load_rest_json(void) { return g_rest.c_str(); }

kct_status kct_save(kct_table *t, const char *path, const char *tail_json) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    if (!path || !tail_json) { set_err("null argument"); return KCT_ERR_ARG; }
    u64 n = 0;
    KCT_TRY(kct_len(t, &n));
    std::vector<u64> keys(n ? n : 1), counts(n ? n : 1);
    u64 got = 0;
    KCT_TRY(kct_dump(t, keys.data(), counts.data(), n, 1, &got));  // by hash: the file is reproducible
    FILE *f = fopen(path, "wb");
    if (!f) { set_err("cannot create %s: %s", path, strerror(errno)); return KCT_ERR_ARG; }
    static const unsigned char header[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4 /* fastest */, 255 /* unknown OS */};
    bool ok = write_all(f, header, sizeof header);
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nthreads = std::max<size_t>(1, std::min<size_t>(8, hw ? hw : 1));
    const size_t pairs_per_piece = (size_t)1 << 16;  // ~1.7 MB of text
    std::vector<z_stream> zs(nthreads);
    for (auto &z : zs) { memset(&z, 0, sizeof z); if (deflateInit2(&z, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) ok = false; }
    std::vector<Piece> pieces(nthreads);
    std::vector<std::string> texts(nthreads);
    uLong crc = crc32(0L, Z_NULL, 0);
    u64 total_len = 0;
    const size_t npieces = std::max<size_t>(1, (got + pairs_per_piece - 1) / pairs_per_piece);
    for (size_t p0 = 0; ok && p0 < npieces; p0 += nthreads) {
        const size_t np = std::min(nthreads, npieces - p0);
        auto work = [&](size_t j) {
            const size_t pi = p0 + j, lo = pi * pairs_per_piece, hi = std::min<size_t>(got, lo + pairs_per_piece);
            std::string &s = texts[j];
            s.clear();
            s.reserve((hi - lo) * 44 + 64 + (pi + 1 == npieces ? strlen(tail_json) : 0));
            if (pi == 0) s += "{\"counts\":{";
            char buf[48];
            for (size_t i = lo; i < hi; ++i) {
                char *q = buf;
                if (i) *q++ = ',';
                *q++ = '"'; q = put_u64(q, keys[i]); *q++ = '"'; *q++ = ':'; q = put_u64(q, counts[i]);
                s.append(buf, (size_t)(q - buf));
            }
            if (pi + 1 == npieces) { s += "}"; s += tail_json; }
            deflate_piece(zs[j], s, pieces[j]);
        };
        std::vector<std::thread> pool;
        for (size_t j = 1; j < np; ++j) pool.emplace_back(work, j);
        work(0);
        for (auto &th : pool) th.join();
        for (size_t j = 0; ok && j < np; ++j) {
            if (pieces[j].err != Z_OK) { ok = false; break; }
            ok = write_all(f, pieces[j].out.data(), pieces[j].out.size());
            crc = crc32_combine(crc, pieces[j].crc, (z_off_t)pieces[j].text_len);
            total_len += pieces[j].text_len;
        }
    }
    for (auto &z : zs) deflateEnd(&z);
    static const unsigned char final_block[2] = {0x03, 0x00};  // an empty fixed-Huffman block with BFINAL set
    unsigned char trailer[8];
    for (int i = 0; i < 4; ++i) { trailer[i] = (unsigned char)(crc >> (8 * i)); trailer[4 + i] = (unsigned char)(total_len >> (8 * i)); }
    ok = ok && write_all(f, final_block, 2) && write_all(f, trailer, 8);
    if (fclose(f) != 0) ok = false;
    if (!ok) { set_err("writing %s failed", path); return KCT_ERR_ARG; }
    return KCT_OK;
}

kct_status kct_load(const char *path, int device, kct_table **out) {
    if (!path || !out) { set_err("null argument"); return KCT_ERR_ARG; }
    *out = nullptr;
    gzFile f = gzopen(path, "rb");  // transparent for plain files
    if (!f) { set_err("cannot open %s", path); return KCT_ERR_ARG; }
    gzbuffer(f, 1 << 20);
    std::vector<char> text;
    size_t used = 0;
    for (;;) {
        if (text.size() - used < ((size_t)1 << 22)) text.resize(std::max<size_t>((size_t)1 << 24, text.size() * 2));
        const int got = gzread(f, text.data() + used, (unsigned)std::min<size_t>(text.size() - used, (size_t)1 << 30));
        if (got < 0) { gzclose(f); set_err("Deserialization error: %s is not readable as gzip or plain text", path); return KCT_ERR_ARG; }
        if (got == 0) break;
        used += (size_t)got;
    }
    gzclose(f);
    Cursor c{text.data(), text.data() + used};
    std::vector<u64> keys, counts;
    std::string rest = "{";
    g_rest.clear();
    u64 ksize = 256;
    auto bad = [&](const char *what) { set_err("Deserialization error: %s at byte %zu", what, (size_t)(c.p - text.data())); return KCT_ERR_ARG; };
    if (!c.eat('{')) return bad("expected '{'");
    if (!c.eat('}')) {
        for (;;) {
            const char *kb, *ke;
            if (!read_string(c, kb, ke)) return bad("expected a member name");
            if (!c.eat(':')) return bad("expected ':'");
            const std::string key(kb, ke);
            if (key == "counts") {
                if (!c.eat('{')) return bad("counts must be an object");
                if (!c.eat('}')) {
                    for (;;) {
                        const char *hb, *he;
                        u64 h, v;
                        if (!read_string(c, hb, he) || !read_u64(hb, he, h)) return bad("counts keys must be decimal u64 strings");
                        if (!c.eat(':')) return bad("expected ':'");
                        c.ws();
                        const char *vb = c.p;
                        while (c.p < c.end && *c.p >= '0' && *c.p <= '9') ++c.p;
                        if (!read_u64(vb, c.p, v)) return bad("counts values must be u64");
                        keys.push_back(h); counts.push_back(v);
                        if (c.eat(',')) continue;
                        if (c.eat('}')) break;
                        return bad("expected ',' or '}' in counts");
                    }
                }
            } else {
                c.ws();
                const char *vb = c.p;
                if (!skip_value(c)) return bad("malformed value");
                if (key == "ksize" && !read_u64(vb, c.p, ksize)) return bad("ksize must be an integer");
                if (rest.size() > 1) rest += ',';
                rest += '"'; rest += key; rest += "\":"; rest.append(vb, c.p);
            }
            if (c.eat(',')) continue;
            if (c.eat('}')) break;
            return bad("expected ',' or '}'");
        }
    }
    rest += '}';
    if (ksize > 255) { set_err("Deserialization error: missing or invalid ksize"); return KCT_ERR_ARG; }
    g_rest = rest;
    kct_table *t = nullptr;
    KCT_TRY(kct_create((uint8_t)ksize, keys.size(), device, &t));
    // serde writes each key once; in a hand-made file with a repeated key the values are summed
    kct_status st = KCT_OK;
    if (!keys.empty()) st = kct_merge_host(t, keys.data(), counts.data(), keys.size(), nullptr, nullptr);
    if (st != KCT_OK) { kct_destroy(t); return st; }
    *out = t;
    return KCT_OK;
}

}  // extern "C"
