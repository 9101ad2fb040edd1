// xml_png.cpp -- the two file formats the host side reads/writes besides OBJ: a minimal XML element tree (what
// load.rs uses of xmltree 0.4) and PNG (png 0.9: load_img load.rs:437-443, save_img main.rs:120-130).  No third-party
// libraries: inflate and the stored-block deflate writer are written out here.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include "hpt_host.h"

namespace hpt_host {

// ------------------------------------------------------------------------------------------------ XML
const std::string* Element::attr(const char* name) const {
    for (const auto& kv : attrs) if (kv.first == name) return &kv.second;
    return nullptr;
}
const Element* Element::child(const char* t) const {
    for (const Element& c : children) if (c.tag == t) return &c;
    return nullptr;
}

namespace {
struct XmlParser {
    const std::string& s; size_t p = 0;
    explicit XmlParser(const std::string& text) : s(text) {}
    [[noreturn]] void fail(const char* what) const { throw Error(std::string("xml: ") + what + " at byte " + std::to_string(p)); }
    bool starts(const char* lit) const { return s.compare(p, std::strlen(lit), lit) == 0; }
    void skip_ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) ++p; }
    void skip_until(const char* lit) {
        const size_t q = s.find(lit, p);
        if (q == std::string::npos) fail("unterminated construct");
        p = q + std::strlen(lit);
    }
    // skips text, comments, processing instructions and DOCTYPE up to the next tag start; false at end of input
    bool next_tag() {
        for (;;) {
            const size_t q = s.find('<', p);
            if (q == std::string::npos) { p = s.size(); return false; }
            p = q;
            if (starts("<!--")) { skip_until("-->"); continue; }
            if (starts("<?")) { skip_until("?>"); continue; }
            if (starts("<![CDATA[")) { skip_until("]]>"); continue; }
            if (starts("<!")) { skip_until(">"); continue; }
            return true;
        }
    }
    static bool name_char(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-' || c == ':' || c == '.'; }
    std::string name() {
        const size_t b = p;
        while (p < s.size() && name_char(s[p])) ++p;
        if (p == b) fail("name expected");
        return s.substr(b, p - b);
    }
    static std::string unescape(const std::string& v) {
        if (v.find('&') == std::string::npos) return v;
        std::string o;
        for (size_t i = 0; i < v.size(); ++i) {
            if (v[i] != '&') { o += v[i]; continue; }
            static const struct { const char* e; char c; } ents[] = {{"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&quot;", '"'}, {"&apos;", '\''}};
            bool done = false;
            for (const auto& en : ents) if (v.compare(i, std::strlen(en.e), en.e) == 0) { o += en.c; i += std::strlen(en.e) - 1; done = true; break; }
            if (!done) o += v[i];
        }
        return o;
    }
    Element element() {   // p at '<' of an opening tag
        ++p;
        Element e; e.tag = name();
        for (;;) {
            skip_ws();
            if (p >= s.size()) fail("unterminated tag");
            if (s[p] == '/') { if (p + 1 >= s.size() || s[p + 1] != '>') fail("'/>' expected"); p += 2; return e; }
            if (s[p] == '>') { ++p; break; }
            std::string k = name();
            skip_ws();
            if (p >= s.size() || s[p] != '=') fail("'=' expected");
            ++p; skip_ws();
            if (p >= s.size() || (s[p] != '"' && s[p] != '\'')) fail("quoted attribute value expected");
            const char q = s[p++];
            const size_t b = p;
            while (p < s.size() && s[p] != q) ++p;
            if (p >= s.size()) fail("unterminated attribute value");
            e.attrs.emplace_back(std::move(k), unescape(s.substr(b, p - b)));
            ++p;
        }
        for (;;) {
            if (!next_tag()) fail("missing closing tag");
            if (starts("</")) {
                p += 2;
                const std::string n = name();
                if (n != e.tag) fail("mismatched closing tag");
                skip_ws();
                if (p >= s.size() || s[p] != '>') fail("'>' expected");
                ++p;
                return e;
            }
            e.children.push_back(element());
        }
    }
};
}   // namespace

Element parse_xml(const std::string& text) {
    XmlParser x(text);
    if (!x.next_tag() || x.starts("</")) throw Error("xml: no root element");
    return x.element();
}

// ------------------------------------------------------------------------------------------------ PNG
namespace {
uint32_t crc_table[256]; bool crc_ready = false;
uint32_t crc32(const uint8_t* d, size_t n, uint32_t crc = 0) {
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; crc_table[i] = c; }
        crc_ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = crc_table[(crc ^ d[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}
uint32_t adler32(const uint8_t* d, size_t n) {
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; ++i) { a = (a + d[i]) % 65521u; b = (b + a) % 65521u; }
    return (b << 16) | a;
}
uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
void put_be32(std::vector<uint8_t>& o, uint32_t v) { o.push_back(v >> 24); o.push_back(v >> 16); o.push_back(v >> 8); o.push_back(v); }

// RFC 1951 inflate (stored, fixed and dynamic Huffman blocks), canonical-code decoding by counting
struct Inflater {
    const uint8_t* in; size_t n, pos = 0; uint32_t bitbuf = 0; int bitcnt = 0;
    std::vector<uint8_t>& out;
    Inflater(const uint8_t* d, size_t len, std::vector<uint8_t>& o) : in(d), n(len), out(o) {}
    int bits(int need) {   // need <= 13
        uint32_t v = bitbuf;
        while (bitcnt < need) {
            if (pos >= n) throw Error("png: truncated deflate stream");
            v |= (uint32_t)in[pos++] << bitcnt; bitcnt += 8;
        }
        bitbuf = v >> need; bitcnt -= need;
        return (int)(v & ((1u << need) - 1u));
    }
    struct Huff { short count[16]; short symbol[288]; };
    static void build(Huff& h, const short* len, int nsym) {
        std::memset(h.count, 0, sizeof h.count);
        for (int s = 0; s < nsym; ++s) h.count[len[s]]++;
        short offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + h.count[l];
        for (int s = 0; s < nsym; ++s) if (len[s]) h.symbol[offs[len[s]]++] = (short)s;
        h.count[0] = 0;
    }
    int decode(const Huff& h) {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; ++l) {
            code |= bits(1);
            const int c = h.count[l];
            if (code - c < first) return h.symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        throw Error("png: bad Huffman code");
    }
    void codes(const Huff& lit, const Huff& dist) {
        static const short lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const short lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const short dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const short dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (;;) {
            int sym = decode(lit);
            if (sym < 256) { out.push_back((uint8_t)sym); continue; }
            if (sym == 256) return;
            sym -= 257;
            if (sym >= 29) throw Error("png: bad length symbol");
            const int len = lbase[sym] + bits(lext[sym]);
            const int ds = decode(dist);
            if (ds >= 30) throw Error("png: bad distance symbol");
            const size_t d = (size_t)dbase[ds] + (size_t)bits(dext[ds]);
            if (d > out.size()) throw Error("png: distance too far back");
            for (int k = 0; k < len; ++k) out.push_back(out[out.size() - d]);
        }
    }
    void run() {
        int last;
        do {
            last = bits(1);
            const int type = bits(2);
            if (type == 0) {
                bitbuf = 0; bitcnt = 0;
                if (pos + 4 > n) throw Error("png: truncated stored block");
                const unsigned len = in[pos] | (in[pos + 1] << 8), nlen = in[pos + 2] | (in[pos + 3] << 8);
                pos += 4;
                if ((len ^ 0xffffu) != nlen || pos + len > n) throw Error("png: bad stored block");
                out.insert(out.end(), in + pos, in + pos + len); pos += len;
            } else if (type == 1) {
                short l[288]; Huff lit, dist;
                for (int s = 0; s < 144; ++s) l[s] = 8;
                for (int s = 144; s < 256; ++s) l[s] = 9;
                for (int s = 256; s < 280; ++s) l[s] = 7;
                for (int s = 280; s < 288; ++s) l[s] = 8;
                build(lit, l, 288);
                for (int s = 0; s < 30; ++s) l[s] = 5;
                build(dist, l, 30);
                codes(lit, dist);
            } else if (type == 2) {
                static const short order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlen = bits(5) + 257, ndist = bits(5) + 1, ncode = bits(4) + 4;
                if (nlen > 286 || ndist > 30) throw Error("png: bad dynamic block header");
                short l[320]; Huff lc, lit, dist;
                for (int i = 0; i < 19; ++i) l[i] = 0;
                for (int i = 0; i < ncode; ++i) l[order[i]] = (short)bits(3);
                build(lc, l, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = decode(lc);
                    if (sym < 16) l[idx++] = (short)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (idx == 0) throw Error("png: repeat without a previous length"); val = l[idx - 1]; rep = 3 + bits(2); }
                        else if (sym == 17) rep = 3 + bits(3);
                        else rep = 11 + bits(7);
                        if (idx + rep > nlen + ndist) throw Error("png: too many code lengths");
                        while (rep--) l[idx++] = (short)val;
                    }
                }
                build(lit, l, nlen);
                build(dist, l + nlen, ndist);
                codes(lit, dist);
            } else throw Error("png: reserved deflate block type");
        } while (!last);
    }
};
}   // namespace

ImageRGB8 read_png_rgb8(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error("cannot open " + path);
    std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) throw Error("not a png: " + path);
    uint32_t w = 0, h = 0; int ctype = -1;
    std::vector<uint8_t> idat;
    size_t pos = 8;
    while (pos + 12 <= d.size()) {
        const uint32_t len = be32(&d[pos]);
        if (pos + 12 + (size_t)len > d.size()) throw Error("png: truncated chunk");
        const uint8_t* body = &d[pos + 8];
        if (std::memcmp(&d[pos + 4], "IHDR", 4) == 0) {
            if (len < 13) throw Error("png: short IHDR");
            w = be32(body); h = be32(body + 4); ctype = body[9];
            if (body[8] != 8 || body[12] != 0 || (ctype != 0 && ctype != 2 && ctype != 4 && ctype != 6)) throw Error("unsupported png format");
        } else if (std::memcmp(&d[pos + 4], "IDAT", 4) == 0) idat.insert(idat.end(), body, body + len);
        else if (std::memcmp(&d[pos + 4], "IEND", 4) == 0) break;
        pos += 12 + (size_t)len;
    }
    if (ctype < 0 || idat.size() < 6) throw Error("png: no image data");
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : 4;
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * ((size_t)w * ch + 1));
    Inflater inf(idat.data() + 2, idat.size() - 2, raw);   // 2-byte zlib header; the Adler-32 trailer is not read
    inf.run();
    const size_t stride = (size_t)w * ch;
    if (raw.size() < (size_t)h * (stride + 1)) throw Error("png: short image data");
    std::vector<uint8_t> cur(stride), prev(stride, 0);
    ImageRGB8 img; img.w = (int32_t)w; img.h = (int32_t)h; img.px.resize((size_t)w * h * 3);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* line = &raw[(size_t)y * (stride + 1)];
        const int ft = line[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
            int pr = 0;
            if (ft == 1) pr = a;
            else if (ft == 2) pr = b;
            else if (ft == 3) pr = (a + b) >> 1;
            else if (ft == 4) { const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c); pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) throw Error("png: bad filter type");
            cur[i] = (uint8_t)((line[1 + i] + pr) & 255);
        }
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t* o = &img.px[((size_t)y * w + x) * 3];
            const uint8_t* s = &cur[(size_t)x * ch];
            if (ch <= 2) { o[0] = o[1] = o[2] = s[0]; } else { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
        }
        prev.swap(cur);
    }
    return img;
}

void write_png_rgba8(const std::string& path, uint32_t w, uint32_t h, const uint8_t* rgba) {
    std::vector<uint8_t> raw; raw.reserve((size_t)h * ((size_t)w * 4 + 1));
    for (uint32_t y = 0; y < h; ++y) { raw.push_back(0); raw.insert(raw.end(), rgba + (size_t)y * w * 4, rgba + (size_t)(y + 1) * w * 4); }
    std::vector<uint8_t> z = {0x78, 0x01};   // zlib header, then stored deflate blocks (valid, uncompressed)
    size_t pos = 0;
    do {
        const size_t len = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + len >= raw.size() ? 1 : 0);
        z.push_back(len & 0xff); z.push_back(len >> 8); z.push_back(~len & 0xff); z.push_back((~len >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + len);
        pos += len;
    } while (pos < raw.size());
    put_be32(z, adler32(raw.data(), raw.size()));
    std::vector<uint8_t> out(std::begin("\x89PNG\r\n\x1a\n"), std::begin("\x89PNG\r\n\x1a\n") + 8);
    auto chunk = [&](const char* typ, const std::vector<uint8_t>& body) {
        put_be32(out, (uint32_t)body.size());
        const size_t start = out.size();
        out.insert(out.end(), typ, typ + 4);
        out.insert(out.end(), body.begin(), body.end());
        put_be32(out, crc32(&out[start], out.size() - start));
    };
    std::vector<uint8_t> ihdr; put_be32(ihdr, w); put_be32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk("IHDR", ihdr); chunk("IDAT", z); chunk("IEND", {});
    std::ofstream f(path, std::ios::binary);
    if (!f) throw Error("cannot create " + path);   // main.rs:122 unwrap
    f.write((const char*)out.data(), (std::streamsize)out.size());
}

}   // namespace hpt_host
