// rosbag.cpp — a dependency-free reader for rosbag format 2.0 files (host side of the C ABI; no ROS, no libbz2, no liblz4).
//
// What it replaces on the reference's side: `rosbag play test_0515.bag` feeding `/lslidar_point_cloud` into the subscriber of
// ImageProjection (README.md:33-37, launch/test2.launch:6-14, src/IP.cpp:106-133, src/imageProjection.cpp:45,49-55) — the bag
// is opened directly, the messages of one topic are handed out in time order, and a sensor_msgs/PointCloud2 message is
// deserialised (ROS1 wire format) and passed through alego_pc2_to_points (pc2.cpp = pcl::fromROSMsg<PointXYZI>).
//
// Format (http://wiki.ros.org/Bags/Format/2.0, restated): "#ROSBAG V2.0\n", then records
//   <header_len u32><header: (<field_len u32><name>=<value>)*><data_len u32><data>, all little endian, header field `op`:
//   0x03 bag header, 0x05 chunk (compression = none | bz2 | lz4, size = uncompressed bytes; data = records 0x02 / 0x07),
//   0x07 connection (conn, topic; data = connection header with type / md5sum / ...), 0x02 message data (conn, time),
//   0x04 index data (ver 1, conn, count; data = count x (time u64, offset u32 into the uncompressed chunk)), 0x06 chunk info.
// Chunks that are not followed by index records (a bag that was never closed) are scanned record by record.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/alego_mi355x.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// bzip2 decompression (the published format: Huffman + MTF + RLE2 + inverse BWT + RLE1, block CRCs checked)
// ---------------------------------------------------------------------------------------------------------------------
struct BitReader {
  const uint8_t* p; size_t n, pos = 0; uint64_t buf = 0; int cnt = 0; bool eof = false;
  BitReader(const uint8_t* d, size_t len) : p(d), n(len) {}
  uint32_t bits(int k) {   // k <= 32, MSB first
    while (cnt < k) { uint64_t b = 0; if (pos < n) b = p[pos++]; else eof = true; buf = (buf << 8) | b; cnt += 8; }
    const uint32_t v = (uint32_t)((buf >> (cnt - k)) & ((k == 32) ? 0xFFFFFFFFull : ((1ull << k) - 1ull)));
    cnt -= k;
    return v;
  }
};

struct BzCrc {
  uint32_t t[256];
  BzCrc() { for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i << 24; for (int k = 0; k < 8; ++k) c = (c & 0x80000000u) ? (c << 1) ^ 0x04c11db7u : (c << 1); t[i] = c; } }
};

// `cap`: the uncompressed size the chunk header declares — output beyond it is refused (an untrusted file must not make the reader allocate without bound)
bool bunzip2(const uint8_t* in, size_t n, std::vector<uint8_t>& out, size_t cap, std::string* err) {
  static const BzCrc crc_tab;
  BitReader br(in, n);
  if (br.bits(8) != 'B' || br.bits(8) != 'Z' || br.bits(8) != 'h') { *err = "bz2: bad stream magic"; return false; }
  const int level = (int)br.bits(8) - '0';
  if (level < 1 || level > 9) { *err = "bz2: bad block size"; return false; }
  const uint32_t block_max = 100000u * level;
  std::vector<uint32_t> tt(block_max);
  uint32_t combined = 0;
  while (true) {
    const uint32_t m1 = br.bits(24), m2 = br.bits(24);
    if (m1 == 0x177245 && m2 == 0x385090) {
      const uint32_t want = br.bits(32);
      if (want != combined) { *err = "bz2: stream CRC mismatch"; return false; }
      return !br.eof || (*err = "bz2: truncated", false);
    }
    if (m1 != 0x314159 || m2 != 0x265359) { *err = "bz2: bad block magic"; return false; }
    const uint32_t block_crc = br.bits(32);
    if (br.bits(1)) { *err = "bz2: randomised blocks are not supported"; return false; }
    const uint32_t orig_ptr = br.bits(24);
    // symbol map
    uint8_t seq2unseq[256];
    int n_in_use = 0;
    const uint32_t used16 = br.bits(16);
    for (int g = 0; g < 16; ++g)
      if (used16 & (0x8000u >> g)) {
        const uint32_t m = br.bits(16);
        for (int k = 0; k < 16; ++k) if (m & (0x8000u >> k)) seq2unseq[n_in_use++] = (uint8_t)(g * 16 + k);
      }
    if (n_in_use == 0) { *err = "bz2: empty symbol map"; return false; }
    const int alpha = n_in_use + 2;
    const int n_groups = (int)br.bits(3);
    if (n_groups < 2 || n_groups > 6) { *err = "bz2: bad group count"; return false; }
    const int n_sel = (int)br.bits(15);
    if (n_sel < 1) { *err = "bz2: no selectors"; return false; }
    std::vector<uint8_t> selector(n_sel);
    {
      uint8_t pos[6] = {0, 1, 2, 3, 4, 5};
      for (int i = 0; i < n_sel; ++i) {
        int j = 0;
        while (br.bits(1)) { if (++j >= n_groups) { *err = "bz2: bad selector"; return false; } }
        const uint8_t v = pos[j];
        for (; j > 0; --j) pos[j] = pos[j - 1];
        pos[0] = v;
        selector[i] = v;
      }
    }
    // code lengths -> canonical decode tables
    int limit[6][22], base[6][22], perm[6][258], min_len[6];
    for (int t = 0; t < n_groups; ++t) {
      uint8_t len[258];
      int cur = (int)br.bits(5);
      for (int s = 0; s < alpha; ++s) {
        while (true) {
          if (cur < 1 || cur > 20) { *err = "bz2: bad code length"; return false; }
          if (!br.bits(1)) break;
          cur += br.bits(1) ? -1 : 1;
        }
        len[s] = (uint8_t)cur;
      }
      int mn = 32, mx = 0;
      for (int s = 0; s < alpha; ++s) { mn = std::min(mn, (int)len[s]); mx = std::max(mx, (int)len[s]); }
      min_len[t] = mn;
      int pp = 0;
      for (int l = mn; l <= mx; ++l) for (int s = 0; s < alpha; ++s) if (len[s] == l) perm[t][pp++] = s;
      int count[22] = {0};
      for (int s = 0; s < alpha; ++s) count[len[s]]++;
      int code = 0, idx = 0;
      for (int l = 0; l < 22; ++l) { limit[t][l] = -1; base[t][l] = 0; }
      for (int l = mn; l <= mx; ++l) {
        base[t][l] = idx - code;          // symbol index = code value + base
        code += count[l]; idx += count[l];
        limit[t][l] = code - 1;           // largest code value of this length
        code <<= 1;
      }
      for (int l = mx + 1; l < 22; ++l) limit[t][l] = 0x7fffffff;
    }
    // MTF + RLE2 decode
    uint8_t mtf[256];
    for (int i = 0; i < 256; ++i) mtf[i] = (uint8_t)i;
    uint32_t nblock = 0;
    int unzftab[256] = {0};
    const int EOB = n_in_use + 1;
    int grp_no = -1, grp_pos = 0, t = 0;
    long run = -1, run_n = 1;
    auto flush_run = [&]() -> bool {
      if (run < 0) return true;
      const uint8_t ch = seq2unseq[mtf[0]];
      const long cnt = run + 1;
      if (nblock + (uint64_t)cnt > block_max) return false;
      unzftab[ch] += (int)cnt;
      for (long k = 0; k < cnt; ++k) tt[nblock++] = ch;
      run = -1; run_n = 1;
      return true;
    };
    while (true) {
      if (grp_pos == 0) { if (++grp_no >= n_sel) { *err = "bz2: selectors exhausted"; return false; } grp_pos = 50; t = selector[grp_no]; }
      --grp_pos;
      int zn = min_len[t];
      int zvec = (int)br.bits(zn);
      while (zvec > limit[t][zn]) { if (++zn > 20) { *err = "bz2: bad Huffman code"; return false; } zvec = (zvec << 1) | (int)br.bits(1); }
      const int pi = zvec + base[t][zn];
      if (pi < 0 || pi >= alpha) { *err = "bz2: bad Huffman symbol"; return false; }
      const int sym = perm[t][pi];
      if (br.eof) { *err = "bz2: truncated block"; return false; }
      if (sym == 0 || sym == 1) {   // RUNA / RUNB
        if (run < 0) { run = -1; run_n = 1; }
        run += (sym == 0 ? 1 : 2) * run_n;
        run_n <<= 1;
        if (run_n > (1L << 21)) { *err = "bz2: run too long"; return false; }
        continue;
      }
      if (!flush_run()) { *err = "bz2: block overflow"; return false; }
      if (sym == EOB) break;
      const int idx = sym - 1;
      const uint8_t v = mtf[idx];
      for (int k = idx; k > 0; --k) mtf[k] = mtf[k - 1];
      mtf[0] = v;
      if (nblock >= block_max) { *err = "bz2: block overflow"; return false; }
      const uint8_t ch = seq2unseq[v];
      unzftab[ch]++;
      tt[nblock++] = ch;
    }
    if (orig_ptr >= nblock) { *err = "bz2: bad BWT origin"; return false; }
    // inverse BWT
    int cftab[257];
    cftab[0] = 0;
    for (int i = 0; i < 256; ++i) cftab[i + 1] = cftab[i] + unzftab[i];
    for (uint32_t i = 0; i < nblock; ++i) { const uint8_t ch = (uint8_t)(tt[i] & 0xff); tt[cftab[ch]++] |= i << 8; }
    uint32_t tpos = tt[orig_ptr] >> 8;
    // RLE1 decode + CRC
    uint32_t crc = 0xffffffffu;
    int same = 0, prev = -1;
    for (uint32_t i = 0; i < nblock; ++i) {
      tpos = tt[tpos];
      const uint8_t ch = (uint8_t)(tpos & 0xff);
      tpos >>= 8;
      if (same == 4) {   // count byte after four equal bytes
        if (out.size() + ch > cap) { *err = "bz2: output exceeds the chunk's declared size"; return false; }
        for (int k = 0; k < ch; ++k) { out.push_back((uint8_t)prev); crc = (crc << 8) ^ crc_tab.t[(crc >> 24) ^ (uint8_t)prev]; }
        same = 0; prev = -1;
        continue;
      }
      if (out.size() >= cap) { *err = "bz2: output exceeds the chunk's declared size"; return false; }
      out.push_back(ch);
      crc = (crc << 8) ^ crc_tab.t[(crc >> 24) ^ ch];
      if ((int)ch == prev) ++same; else { same = 1; prev = ch; }
    }
    crc = ~crc;
    if (crc != block_crc) { *err = "bz2: block CRC mismatch"; return false; }
    combined = ((combined << 1) | (combined >> 31)) ^ crc;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LZ4 frame format (what rosbag's "lz4" compression writes); checksums are not verified
// ---------------------------------------------------------------------------------------------------------------------
bool unlz4_frame(const uint8_t* in, size_t n, std::vector<uint8_t>& out, size_t cap, std::string* err) {
  size_t p = 0;
  auto need = [&](size_t k) { return p + k <= n; };
  if (!need(7) || in[0] != 0x04 || in[1] != 0x22 || in[2] != 0x4D || in[3] != 0x18) { *err = "lz4: bad frame magic"; return false; }
  const uint8_t flg = in[4];
  if ((flg >> 6) != 1) { *err = "lz4: unsupported frame version"; return false; }
  p = 6;
  if (flg & 0x08) p += 8;   // content size
  if (flg & 0x01) p += 4;   // dictionary id
  p += 1;                   // header checksum
  const bool block_checksum = flg & 0x10;
  while (true) {
    if (!need(4)) { *err = "lz4: truncated frame"; return false; }
    uint32_t bs; std::memcpy(&bs, in + p, 4); p += 4;
    if (bs == 0) return true;   // end mark (a content checksum may follow)
    const bool raw = bs & 0x80000000u;
    bs &= 0x7fffffffu;
    if (!need(bs)) { *err = "lz4: truncated block"; return false; }
    if (raw) { if (out.size() + bs > cap) { *err = "lz4: output exceeds the chunk's declared size"; return false; } out.insert(out.end(), in + p, in + p + bs); }
    else {
      size_t q = p; const size_t end = p + bs;
      while (q < end) {
        const uint8_t tok = in[q++];
        size_t lit = tok >> 4;
        if (lit == 15) { uint8_t b; do { if (q >= end) { *err = "lz4: bad literal length"; return false; } b = in[q++]; lit += b; } while (b == 255); }
        if (q + lit > end) { *err = "lz4: literals beyond the block"; return false; }
        if (out.size() + lit > cap) { *err = "lz4: output exceeds the chunk's declared size"; return false; }
        out.insert(out.end(), in + q, in + q + lit); q += lit;
        if (q >= end) break;   // the last sequence has no match
        if (q + 2 > end) { *err = "lz4: truncated match"; return false; }
        const size_t off = in[q] | ((size_t)in[q + 1] << 8); q += 2;
        size_t ml = (tok & 15);
        if (ml == 15) { uint8_t b; do { if (q >= end) { *err = "lz4: bad match length"; return false; } b = in[q++]; ml += b; } while (b == 255); }
        ml += 4;
        if (off == 0 || off > out.size()) { *err = "lz4: bad match offset"; return false; }
        if (out.size() + ml > cap) { *err = "lz4: output exceeds the chunk's declared size"; return false; }
        const size_t s = out.size() - off;
        for (size_t k = 0; k < ml; ++k) out.push_back(out[s + k]);   // (overlapping copies are the format's run-length idiom)
      }
    }
    p += bs;
    if (block_checksum) p += 4;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// records
// ---------------------------------------------------------------------------------------------------------------------
struct Field { const uint8_t* v; uint32_t n; };
using Fields = std::map<std::string, Field>;

inline uint32_t rd32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

// <len u32><name=value>... of `hl` bytes
bool parse_fields(const uint8_t* h, uint32_t hl, Fields& f) {
  f.clear();
  uint32_t p = 0;
  while (p < hl) {
    if (p + 4 > hl) return false;
    const uint32_t fl = rd32(h + p); p += 4;
    if (fl > hl - p) return false;
    const uint8_t* eq = (const uint8_t*)std::memchr(h + p, '=', fl);
    if (!eq) return false;
    f[std::string((const char*)h + p, eq - (h + p))] = Field{eq + 1, (uint32_t)(fl - (eq + 1 - (h + p)))};
    p += fl;
  }
  return true;
}
bool f_u32(const Fields& f, const char* k, uint32_t* v) { auto it = f.find(k); if (it == f.end() || it->second.n != 4) return false; *v = rd32(it->second.v); return true; }
bool f_u64(const Fields& f, const char* k, uint64_t* v) { auto it = f.find(k); if (it == f.end() || it->second.n != 8) return false; *v = rd64(it->second.v); return true; }
std::string f_str(const Fields& f, const char* k) { auto it = f.find(k); return it == f.end() ? std::string() : std::string((const char*)it->second.v, it->second.n); }
int f_op(const Fields& f) { auto it = f.find("op"); return (it == f.end() || it->second.n != 1) ? -1 : it->second.v[0]; }

struct Record { Fields h; const uint8_t* data; uint32_t dlen; size_t next; };
// one record at buf[pos..]; false = truncated / malformed
bool read_record(const uint8_t* buf, size_t len, size_t pos, Record& r) {
  if (pos + 4 > len) return false;
  const uint32_t hl = rd32(buf + pos);
  if (hl > len - pos - 4) return false;
  if (!parse_fields(buf + pos + 4, hl, r.h)) return false;
  const size_t dp = pos + 4 + hl;
  if (dp + 4 > len) return false;
  r.dlen = rd32(buf + dp);
  if (r.dlen > len - dp - 4) return false;
  r.data = buf + dp + 4;
  r.next = dp + 4 + r.dlen;
  return true;
}

struct Conn { std::string topic, type, md5; };
struct Chunk { size_t pos; uint32_t clen, size; int comp; bool indexed; };   // comp 0 none, 1 bz2, 2 lz4
struct Msg { uint64_t time_ns; uint32_t chunk, offset; uint64_t seq; };

}  // namespace

struct alego_bag {
  int fd = -1;
  const uint8_t* base = nullptr;
  size_t len = 0;
  std::map<uint32_t, Conn> conns;
  std::vector<Chunk> chunks;
  std::map<std::string, std::vector<Msg>> topics;    // messages per topic in (time, file order)
  std::vector<std::string> topic_names;
  std::map<std::string, std::string> topic_type;
  int cached = -1;
  std::vector<uint8_t> cache;
  std::vector<alego_pc2_field> fields_tmp;
  std::vector<std::string> names_tmp;
  std::string err;

  const uint8_t* chunk_bytes(uint32_t ci, size_t* n) {
    const Chunk& c = chunks[ci];
    if (c.comp == 0) { *n = c.clen; return base + c.pos; }
    if (cached != (int)ci) {
      cache.clear(); cache.reserve(std::min<size_t>(c.size, (size_t)64 << 20));   // (header-controlled: grown on demand beyond 64 MiB, never past c.size)
      cached = -1;
      const bool ok = c.comp == 1 ? bunzip2(base + c.pos, c.clen, cache, c.size, &err) : unlz4_frame(base + c.pos, c.clen, cache, c.size, &err);
      if (!ok) return nullptr;
      if (cache.size() != c.size) { err = "chunk: uncompressed size differs from the header's"; return nullptr; }
      cached = (int)ci;
    }
    *n = cache.size();
    return cache.data();
  }
};

namespace {

void add_conn(alego_bag* b, const Record& r) {
  uint32_t id;
  if (!f_u32(r.h, "conn", &id)) return;
  Conn c;
  c.topic = f_str(r.h, "topic");
  Fields ch;
  if (parse_fields(r.data, r.dlen, ch)) { c.type = f_str(ch, "type"); c.md5 = f_str(ch, "md5sum"); const std::string t = f_str(ch, "topic"); if (c.topic.empty()) c.topic = t; }
  b->conns[id] = c;
}

// every record of an (uncompressed) chunk: connections, and — for chunks without index records — the messages
bool scan_chunk(alego_bag* b, uint32_t ci, bool want_msgs, std::vector<std::pair<uint32_t, Msg>>* found) {
  size_t n = 0;
  const uint8_t* d = b->chunk_bytes(ci, &n);
  if (!d) return false;
  size_t pos = 0;
  Record r;
  while (pos < n) {
    if (!read_record(d, n, pos, r)) { b->err = "chunk: malformed record"; return false; }
    const int op = f_op(r.h);
    if (op == 0x07) add_conn(b, r);
    else if (op == 0x02 && want_msgs) {
      uint32_t conn; uint64_t t;
      if (f_u32(r.h, "conn", &conn) && f_u64(r.h, "time", &t))
        found->push_back({conn, Msg{(t & 0xffffffffull) * 1000000000ull + (t >> 32), ci, (uint32_t)pos, 0}});
    }
    pos = r.next;
  }
  return true;
}

}  // namespace

extern "C" {

const char* alego_bag_last_error(const alego_bag* b) { return b ? b->err.c_str() : "null bag"; }

void alego_bag_close(alego_bag* b) {
  if (!b) return;
  if (b->base) munmap(const_cast<uint8_t*>(b->base), b->len);
  if (b->fd >= 0) close(b->fd);
  delete b;
}

static int bag_open_impl(const char* path, alego_bag** out, alego_bag*& b) {
  b = new alego_bag();
  auto fail = [&](const char* why) { std::fprintf(stderr, "alego_bag_open(%s): %s%s%s\n", path, why, b->err.empty() ? "" : ": ", b->err.c_str()); alego_bag_close(b); return ALEGO_ERR_ARG; };
  b->fd = open(path, O_RDONLY);
  if (b->fd < 0) return fail("cannot open the file");
  struct stat st;
  if (fstat(b->fd, &st) != 0 || st.st_size < 13) return fail("not a rosbag");
  b->len = (size_t)st.st_size;
  void* m = mmap(nullptr, b->len, PROT_READ, MAP_PRIVATE, b->fd, 0);
  if (m == MAP_FAILED) { b->base = nullptr; return fail("mmap failed"); }
  b->base = (const uint8_t*)m;
  if (std::memcmp(b->base, "#ROSBAG V2.0\n", 13) != 0) return fail("not a rosbag format 2.0 file (\"#ROSBAG V2.0\" expected)");
  std::vector<std::pair<uint32_t, Msg>> found;   // (conn, message)
  size_t pos = 13;
  Record r;
  while (pos < b->len) {
    if (!read_record(b->base, b->len, pos, r)) {
      if (b->chunks.empty()) return fail("malformed record");
      break;   // a truncated tail (recording interrupted): keep what is complete
    }
    const int op = f_op(r.h);
    if (op == 0x05) {
      const std::string comp = f_str(r.h, "compression");
      uint32_t size = 0;
      if (!f_u32(r.h, "size", &size)) return fail("chunk without size");
      const int c = comp == "none" ? 0 : comp == "bz2" ? 1 : comp == "lz4" ? 2 : -1;
      if (c < 0) { b->err = comp; return fail("unknown chunk compression"); }
      b->chunks.push_back(Chunk{(size_t)(r.data - b->base), r.dlen, size, c, false});
    } else if (op == 0x04 && !b->chunks.empty()) {
      uint32_t ver = 0, conn = 0, count = 0;
      if (f_u32(r.h, "ver", &ver) && ver == 1 && f_u32(r.h, "conn", &conn) && f_u32(r.h, "count", &count) && (uint64_t)count * 12 <= r.dlen) {
        for (uint32_t i = 0; i < count; ++i) {
          const uint8_t* e = r.data + (size_t)i * 12;
          found.push_back({conn, Msg{(uint64_t)rd32(e) * 1000000000ull + rd32(e + 4), (uint32_t)b->chunks.size() - 1, rd32(e + 8), 0}});
        }
        b->chunks.back().indexed = true;
      }
    } else if (op == 0x07) {
      add_conn(b, r);
    }
    pos = r.next;
  }
  // chunks to walk record by record: those without index records, and those an index entry of which names a connection no record
  // outside the chunks declared (one pass over the index entries, not one per chunk)
  std::vector<uint8_t> needs_scan(b->chunks.size(), 0);
  for (uint32_t ci = 0; ci < b->chunks.size(); ++ci) needs_scan[ci] = !b->chunks[ci].indexed;
  for (auto& f : found) if (!needs_scan[f.second.chunk] && !b->conns.count(f.first)) needs_scan[f.second.chunk] = 1;
  for (uint32_t ci = 0; ci < b->chunks.size(); ++ci)
    if (needs_scan[ci] && !scan_chunk(b, ci, !b->chunks[ci].indexed, &found)) return fail("unreadable chunk");
  uint64_t seq = 0;
  for (auto& f : found) {
    auto it = b->conns.find(f.first);
    if (it == b->conns.end()) continue;
    f.second.seq = seq++;
    b->topics[it->second.topic].push_back(f.second);
    b->topic_type[it->second.topic] = it->second.type;
  }
  for (auto& kv : b->topics) {
    std::stable_sort(kv.second.begin(), kv.second.end(), [](const Msg& a, const Msg& c) { return a.time_ns != c.time_ns ? a.time_ns < c.time_ns : (a.chunk != c.chunk ? a.chunk < c.chunk : a.offset < c.offset); });
    b->topic_names.push_back(kv.first);
  }
  *out = b;
  return ALEGO_OK;
}

int alego_bag_open(const char* path, alego_bag** out) {
  if (!path || !out) return ALEGO_ERR_ARG;
  *out = nullptr;
  alego_bag* b = nullptr;
  // nothing may unwind through the C ABI: allocation failures on a hostile or damaged file become an error code
  try { return bag_open_impl(path, out, b); }
  catch (...) { std::fprintf(stderr, "alego_bag_open(%s): out of memory or malformed file\n", path); if (b && *out != b) alego_bag_close(b); *out = nullptr; return ALEGO_ERR_ARG; }
}

int alego_bag_topic_count(const alego_bag* b) { return b ? (int)b->topic_names.size() : ALEGO_ERR_ARG; }

int alego_bag_topic_info(const alego_bag* b, int i, const char** topic, const char** datatype, int64_t* n_messages) {
  if (!b || i < 0 || i >= (int)b->topic_names.size()) return ALEGO_ERR_ARG;
  const std::string& t = b->topic_names[i];
  if (topic) *topic = t.c_str();
  if (datatype) *datatype = b->topic_type.at(t).c_str();
  if (n_messages) *n_messages = (int64_t)b->topics.at(t).size();
  return ALEGO_OK;
}

int64_t alego_bag_message_count(const alego_bag* b, const char* topic) {
  if (!b || !topic) return ALEGO_ERR_ARG;
  auto it = b->topics.find(topic);
  return it == b->topics.end() ? 0 : (int64_t)it->second.size();
}

static int bag_read_raw_impl(alego_bag* b, const char* topic, int64_t index, const uint8_t** data, uint64_t* len, double* bag_time) {
  auto it = b->topics.find(topic);
  if (it == b->topics.end() || index < 0 || index >= (int64_t)it->second.size()) { b->err = "no such topic / message index"; return ALEGO_ERR_ARG; }
  const Msg& m = it->second[(size_t)index];
  size_t n = 0;
  const uint8_t* d = b->chunk_bytes(m.chunk, &n);
  if (!d) return ALEGO_ERR_ARG;
  Record r;
  if (m.offset >= n || !read_record(d, n, m.offset, r) || f_op(r.h) != 0x02) { b->err = "index entry does not point at a message record"; return ALEGO_ERR_ARG; }
  *data = r.data; *len = r.dlen;   // valid until the next read of another compressed chunk / alego_bag_close
  if (bag_time) *bag_time = (double)(m.time_ns / 1000000000ull) + 1e-9 * (double)(m.time_ns % 1000000000ull);
  return ALEGO_OK;
}

int alego_bag_read_raw(alego_bag* b, const char* topic, int64_t index, const uint8_t** data, uint64_t* len, double* bag_time) {
  if (!b || !topic || !data || !len) return ALEGO_ERR_ARG;
  try { return bag_read_raw_impl(b, topic, index, data, len, bag_time); }
  catch (...) { b->cached = -1; b->err = "out of memory while reading a chunk"; return ALEGO_ERR_ARG; }
}

// sensor_msgs/PointCloud2 in ROS1 serialisation: Header (seq u32, stamp sec u32 nsec u32, frame_id string), height u32, width u32,
// PointField[] (name string, offset u32, datatype u8, count u32), is_bigendian u8, point_step u32, row_step u32, data u8[], is_dense u8
static int bag_read_pc2_impl(alego_bag* b, const char* topic, int64_t index, alego_point* out, int32_t cap, double* header_stamp, int32_t* is_dense) {
  const uint8_t* d = nullptr; uint64_t n = 0;
  if (int rc = alego_bag_read_raw(b, topic, index, &d, &n, nullptr)) return rc;
  uint64_t p = 0;
  bool ok = true;
  auto u32 = [&]() -> uint32_t { if (p + 4 > n) { ok = false; return 0; } const uint32_t v = rd32(d + p); p += 4; return v; };
  auto u8 = [&]() -> uint8_t { if (p + 1 > n) { ok = false; return 0; } return d[p++]; };
  auto str = [&]() -> std::string { const uint32_t l = u32(); if (!ok || p + l > n) { ok = false; return std::string(); } std::string s((const char*)d + p, l); p += l; return s; };
  (void)u32();
  const uint32_t sec = u32(), nsec = u32();
  (void)str();
  const uint32_t height = u32(), width = u32(), nf = u32();
  if (!ok || nf > 1024) { b->err = "not a sensor_msgs/PointCloud2 message"; return ALEGO_ERR_ARG; }
  b->names_tmp.resize(nf); b->fields_tmp.resize(nf);
  for (uint32_t i = 0; i < nf; ++i) {
    b->names_tmp[i] = str();
    const uint32_t off = u32(); const uint8_t dt = u8(); const uint32_t cnt = u32();
    b->fields_tmp[i] = alego_pc2_field{nullptr, off, dt, cnt};
  }
  for (uint32_t i = 0; i < nf; ++i) b->fields_tmp[i].name = b->names_tmp[i].c_str();
  const uint8_t big = u8();
  const uint32_t point_step = u32(), row_step = u32(), dlen = u32();
  if (!ok || p + dlen + 1 > n) { b->err = "truncated sensor_msgs/PointCloud2 message"; return ALEGO_ERR_ARG; }
  const uint8_t* payload = d + p;
  const uint8_t dense = d[p + dlen];
  if (header_stamp) *header_stamp = (double)sec + 1e-9 * (double)nsec;
  if (is_dense) *is_dense = dense ? 1 : 0;
  const int rc = alego_pc2_to_points(payload, dlen, width, height, point_step, row_step, big, b->fields_tmp.data(), (int)nf, out, cap);
  if (rc < 0) b->err = rc == ALEGO_ERR_CAPACITY ? "cloud larger than the output capacity" : "PointCloud2 layout rejected (x / y / z FLOAT32 fields, steps, data length)";
  return rc;
}

int alego_bag_read_pc2(alego_bag* b, const char* topic, int64_t index, alego_point* out, int32_t cap, double* header_stamp, int32_t* is_dense) {
  if (!b) return ALEGO_ERR_ARG;
  try { return bag_read_pc2_impl(b, topic, index, out, cap, header_stamp, is_dense); }
  catch (...) { b->err = "out of memory while parsing a PointCloud2 message"; return ALEGO_ERR_ARG; }
}

}  // extern "C"
