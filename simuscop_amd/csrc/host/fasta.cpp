// host/fasta.cpp -- see fasta.h.
#include "fasta.h"

#include <chrono>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

#include "../../../include/simuscop_amd.h"
#include "common.h"
#include "reader_pool.h"

namespace simu {

static std::string plain_path(const std::string& ref_file) {
  std::string path = ref_file;
  if (path.empty()) throw Error("genome sequence file not specified!");
  if (path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0) {
    // same side effect as the reference: decompress next to the archive
    std::string plain = path.substr(0, path.size() - 3);
    std::string cmd = "gzip -cd " + path + " > " + plain;
    if (system(cmd.c_str()) != 0) throw Error("could not decompress " + path);
    path = plain;
  }
  return path;
}

static std::string header_key(const std::string& header, bool crlf_as_lf) {  // Fasta.cpp:58-69
  size_t b = header.find_first_not_of(" \t");
  std::string tok;
  if (b != std::string::npos) {
    // (the reference cuts the name at blanks and tabs only: the carriage return of a CR LF file stays in it)
    size_t e = header.find_first_of(crlf_as_lf ? " \t\r" : " \t", b);
    tok = header.substr(b, e == std::string::npos ? std::string::npos : e - b);
  }
  return abbr_of_chr(tok);
}

namespace {
struct EngineError {};
void eng_check(sg_ctx* ctx, int rc, const char* what) {
  if (rc != SG_OK) throw Error(std::string("GPU engine error in ") + what + ": " + sg_last_error(ctx));
}
std::string pread_string(int fd, uint64_t off, uint64_t n) {
  std::string s((size_t)n, '\0');
  uint64_t have = 0;
  while (have < n) {
    ssize_t got = pread(fd, &s[have], n - have, (off_t)(off + have));
    if (got <= 0) break;
    have += (uint64_t)got;
  }
  s.resize((size_t)have);
  return s;
}

// Index rows (what a .fai line holds) of the contigs whose header lines start at the sorted file offsets `hdr`:
// a few small preads per contig -- header text, first line (bases / bytes per line), last bytes.  False when a
// contig's body cannot be a run of equal-width lines.
bool rows_from_headers(int fd, uint64_t size, const std::vector<uint64_t>& hdr, std::vector<std::string>& keys,
                       std::vector<FastaContig>& rows, bool crlf_as_lf) {
  for (size_t i = 0; i < hdr.size(); i++) {
    const uint64_t region_end = i + 1 < hdr.size() ? hdr[i + 1] : size;
    // header line
    std::string head;
    uint64_t p = hdr[i] + 1, first = region_end;
    for (;;) {
      const std::string blk = pread_string(fd, p, std::min<uint64_t>(4096, region_end - p));
      if (blk.empty()) break;
      const size_t nl = blk.find('\n');
      if (nl != std::string::npos) { head.append(blk, 0, nl); first = p + nl + 1; break; }
      head += blk;
      p += blk.size();
    }
    FastaContig row;
    row.raw_offset = first;
    // trailing line breaks (and blank lines) of the region do not belong to a line
    uint64_t end = region_end;
    while (end > first) {
      const uint64_t n = std::min<uint64_t>(64, end - first);
      const std::string tail = pread_string(fd, end - n, n);
      size_t k = tail.size();
      while (k > 0 && (tail[k - 1] == '\n' || tail[k - 1] == '\r')) k--;
      end -= tail.size() - k;
      if (k > 0) break;
    }
    const uint64_t body = end - first;
    if (body) {
      // first line: bases per line, bytes per line
      uint64_t q = first, lb = body, lw = body;
      bool found_nl = false;
      while (q < end && !found_nl) {
        const std::string blk = pread_string(fd, q, std::min<uint64_t>(1u << 16, end - q));
        if (blk.empty()) break;
        const size_t nl = blk.find('\n');
        if (nl != std::string::npos) {
          const uint64_t at = q + nl;  // file offset of the '\n'
          const bool cr = at > first && (nl > 0 ? blk[nl - 1] == '\r' : pread_string(fd, at - 1, 1) == "\r");
          // carriage returns that stay bases (the reference's reading): not a file of fixed-width lines of bases only --
          // the general parser takes it
          if (cr && !crlf_as_lf) return false;
          lb = at - first - (cr ? 1 : 0);
          lw = at - first + 1;
          found_nl = true;
        }
        q += blk.size();
      }
      if (lb == 0 || lb > 0xFFFFFFF0ull || lw > 0xFFFFFFF0ull) return false;
      const uint64_t k = body / lw, r = body % lw;
      if (r > lb) return false;
      row.length = k * lb + r;
      row.line_bases = (uint32_t)lb;
      row.line_width = (uint32_t)lw;
    } else {
      row.line_bases = row.line_width = 1;
    }
    if (!crlf_as_lf && head.find('\r') != std::string::npos) return false;
    keys.push_back(header_key(head, crlf_as_lf));
    rows.push_back(row);
  }
  return true;
}
}  // namespace

bool Fasta::note_name(const std::string& key, bool seen) {
  if (!seen) { names.push_back(key); return false; }
  if (unique_contigs) throw Error("ERROR: contig name " + key + " stands more than once in the reference sequence file (--unique-contigs)");
  names.insert(std::find(names.begin(), names.end(), key) + 1, key);   // next to its first place (the index is sorted by offset)
  return true;
}

void Fasta::open_on_device(const std::string& ref_file, sg_ctx* ctx, int threads) {
  const std::string path = plain_path(ref_file);
  on_device = true;
  names.clear(); seqs.clear(); contigs.clear(); contig_of.clear();
  int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) throw Error("could not open " + path);
  struct stat sb;
  if (fstat(fd, &sb) != 0) { ::close(fd); throw Error("could not open " + path); }
  const uint64_t size = (uint64_t)sb.st_size;
  bool ok = false;
  void* stage[2] = {nullptr, nullptr};
  try {
    // ---- 1. stream the file image to HBM: reader threads fill one pinned buffer while the other copies
    const bool trace = getenv("SIMU_TRACE_LOAD") != nullptr;   // phase times of the ingest on stderr
    using Clk = std::chrono::steady_clock;
    auto secs = [](Clk::time_point a) { return std::chrono::duration<double>(Clk::now() - a).count(); };
    auto t_ph = Clk::now();
    eng_check(ctx, sg_reference_begin(ctx, size), "sg_reference_begin");
    const double t_begin = secs(t_ph);
    t_ph = Clk::now();
    const uint64_t kChunk = 64u << 20;
    for (int i = 0; i < 2 && size; i++) {
      eng_check(ctx, sg_host_alloc(ctx, std::min<uint64_t>(kChunk, size), &stage[i]), "sg_host_alloc");
    }
    const double t_pin = secs(t_ph);
    double t_read = 0, t_wait = 0;
    t_ph = Clk::now();
    int cur = 0;
    ReaderPool pool((int)std::min<uint64_t>((uint64_t)std::max(1, threads), 16));   // (16 readers of 4 MB fill a 64 MB chunk)
    for (uint64_t off = 0; off < size; off += kChunk, cur ^= 1) {
      const uint64_t n = std::min<uint64_t>(kChunk, size - off);
      auto t1 = Clk::now();
      parallel_pread(pool, fd, (uint8_t*)stage[cur], off, n);  // overlaps the copy of the other buffer
      t_read += secs(t1);
      t1 = Clk::now();
      eng_check(ctx, sg_sync(ctx), "sg_sync");                    // the other buffer's copy is done: it is free next round
      t_wait += secs(t1);
      eng_check(ctx, sg_reference_chunk(ctx, off, stage[cur], n), "sg_reference_chunk");
    }
    eng_check(ctx, sg_sync(ctx), "sg_sync");
    if (trace)
      fprintf(stderr, "load trace: device buffer %.3fs, pinned staging %.3fs, stream %.3fs (pread %.3fs, waiting for copies %.3fs) of %.2f GB\n",
              t_begin, t_pin, secs(t_ph), t_read, t_wait, size / 1e9);
    t_ph = Clk::now();
    // ---- 2. header offsets from the device scan; header text and line shape from a few small preads
    std::vector<uint64_t> hdr(1u << 16);
    uint32_t found = 0, flags = 0;
    eng_check(ctx, sg_reference_scan(ctx, hdr.data(), (uint32_t)hdr.size(), &found, &flags), "sg_reference_scan");
    if (found > hdr.size()) {
      hdr.resize(found);
      eng_check(ctx, sg_reference_scan(ctx, hdr.data(), (uint32_t)hdr.size(), &found, &flags), "sg_reference_scan");
    }
    hdr.resize(found);
    std::sort(hdr.begin(), hdr.end());
    bool uniform = !(flags & 1u) && found > 0;
    std::vector<std::string> keys;
    std::vector<FastaContig> rows;
    if (uniform) uniform = rows_from_headers(fd, size, hdr, keys, rows, crlf_as_lf);
    if (uniform) {
      std::vector<sg_contig> tab;
      for (const FastaContig& r : rows) tab.push_back(sg_contig{r.raw_offset, r.length, r.line_bases, r.line_width});
      const int rc = sg_reference_commit(ctx, tab.data(), (uint32_t)tab.size());
      if (rc == SG_ERR_FORMAT) uniform = false;
      else eng_check(ctx, rc, "sg_reference_commit");
    }
    if (uniform) {
      contigs = rows;
      for (size_t i = 0; i < keys.size(); i++) {  // (a repeated name: listed again, resolved to its first sequence)
        if (!note_name(keys[i], contig_of.count(keys[i]) != 0)) contig_of[keys[i]] = (uint32_t)i;
      }
      streamed = true;
      ok = true;
    }
    if (trace) fprintf(stderr, "load trace: header scan, line shapes and ingest kernel %.3fs\n", secs(t_ph));
  } catch (...) {
    ::close(fd);
    for (void* b : stage) if (b) sg_host_free(ctx, b);
    throw;
  }
  ::close(fd);
  for (void* b : stage) if (b) sg_host_free(ctx, b);
  if (!ok) {
    // ---- general parser on the host, result uploaded as one line per contig
    open(ref_file);
    on_device = true;
    streamed = false;
    contigs.clear(); contig_of.clear();  // rebuilt below with the offsets of the uploaded image
    uint64_t total = 0;
    std::vector<sg_contig> tab;
    for (const std::string& k : names) {
      if (contig_of.count(k)) continue;   // (a repeated name: one sequence, listed twice)
      const std::string& s = seqs.at(k);
      contig_of[k] = (uint32_t)contigs.size();
      FastaContig row;
      row.raw_offset = total; row.length = s.size();
      row.line_bases = row.line_width = (uint32_t)std::max<size_t>(1, std::min<size_t>(s.size(), 0xFFFFFFF0u));
      if (s.size() > 0xFFFFFFF0ull) throw Error("ERROR: contig " + k + " is longer than 4 Gbp");
      contigs.push_back(row);
      tab.push_back(sg_contig{row.raw_offset, row.length, row.line_bases, row.line_width});
      total += s.size();
    }
    eng_check(ctx, sg_reference_begin(ctx, total), "sg_reference_begin");
    for (const auto& kv : contig_of) {
      const std::string& s = seqs.at(kv.first);
      eng_check(ctx, sg_reference_chunk(ctx, contigs[kv.second].raw_offset, s.data(), s.size()), "sg_reference_chunk");
    }
    eng_check(ctx, sg_sync(ctx), "sg_sync");
    eng_check(ctx, sg_reference_commit(ctx, tab.data(), (uint32_t)tab.size()), "sg_reference_commit");
    seqs.clear();
  }
  if (names.empty()) throw Error("ERROR: reference sequence cannot be empty!");
}

void Fasta::open(const std::string& ref_file) {
  const std::string path = plain_path(ref_file);
  on_device = false;
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) throw Error("could not open " + path);
  names.clear();
  seqs.clear();
  std::string* cur = nullptr;
  std::string ignored;   // the sequence under a repeated name is never read (note_name)
  std::vector<char> buf(1 << 22);
  std::string header;
  bool in_header = false, line_start = true, skip_line = false;
  size_t got;
  while ((got = fread(buf.data(), 1, buf.size(), fp)) > 0) {
    const char* p = buf.data();
    const char* end = p + got;
    while (p < end) {
      if (in_header || skip_line) {
        const char* nl = (const char*)memchr(p, '\n', end - p);
        if (in_header) header.append(p, (nl ? nl : end) - p);
        if (!nl) break;
        p = nl + 1;
        if (in_header) {
          in_header = false;
          std::string key = header_key(header, crlf_as_lf);
          cur = note_name(key, seqs.count(key) != 0) ? &ignored : &seqs[key];
          cur->clear();
          header.clear();
        }
        skip_line = false;
        line_start = true;
        continue;
      }
      if (line_start) {
        if (*p == '>' || *p == '@') { in_header = true; p++; line_start = false; continue; }
        if (*p == ';') { skip_line = true; p++; line_start = false; continue; }
      }
      const char* nl = (const char*)memchr(p, '\n', end - p);
      const char* stop = nl ? nl : end;
      if (crlf_as_lf && nl && stop > p && stop[-1] == '\r') stop--;  // --crlf-as-lf; by default the carriage return stays, as a base (fasta.h)
      if (cur && stop > p) {
        size_t old = cur->size();
        cur->append(p, stop - p);
        for (size_t i = old; i < cur->size(); i++) {
          char& c = (*cur)[i];
          if (c >= 'a' && c <= 'z') c -= 32;
        }
      }
      if (nl) { p = nl + 1; line_start = true; }
      else { p = end; line_start = false; }
    }
  }
  fclose(fp);
  if (names.empty()) throw Error("ERROR: reference sequence cannot be empty!");
  contigs.clear(); contig_of.clear();
  for (const std::string& k : names) {  // index rows (lengths only: no file offsets in this mode)
    if (contig_of.count(k)) continue;
    contig_of[k] = (uint32_t)contigs.size();
    FastaContig row;
    row.length = seqs.at(k).size();
    contigs.push_back(row);
  }
}

// ---- sharded ingest -------------------------------------------------------------------------------------
bool Fasta::load_index(const std::string& ref_file, int threads) {
  const std::string path = plain_path(ref_file);
  int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) throw Error("could not open " + path);
  struct stat sb;
  if (fstat(fd, &sb) != 0) { ::close(fd); throw Error("could not open " + path); }
  const uint64_t size = (uint64_t)sb.st_size;
  std::vector<std::string> keys;
  std::vector<FastaContig> rows;
  bool ok = false;
  // 1. a .fai next to the file: NAME LENGTH OFFSET LINEBASES LINEWIDTH (Fasta.cpp:45-85)
  struct stat ib;
  const std::string fai = path + ".fai";
  if (stat(fai.c_str(), &ib) == 0 && ib.st_mtime >= sb.st_mtime) {
    if (FILE* f = fopen(fai.c_str(), "r")) {
      char line[4096];
      ok = true;
      while (fgets(line, sizeof line, f)) {
        char name[2048];
        unsigned long long len, off, lb, lw;
        if (sscanf(line, "%2047s %llu %llu %llu %llu", name, &len, &off, &lb, &lw) != 5 || lb == 0 || lw < lb) { ok = false; break; }
        FastaContig r;
        r.length = len; r.raw_offset = off; r.line_bases = (uint32_t)lb; r.line_width = (uint32_t)lw;
        if (off + len + (len ? (len - 1) / lb : 0) * (lw - lb) > size) { ok = false; break; }
        keys.push_back(abbr_of_chr(name));
        rows.push_back(r);
      }
      fclose(f);
      if (rows.empty()) ok = false;
      // An index newer than the file can still be another file's (cp -p, touch, an edit that keeps the size): every row
      // is held against the bytes it points at -- a line break before the sequence, a '>' header with the same key in
      // front of it, and a line break, a header or the end of the file right behind the last base.  Any mismatch: the
      // header scan below decides, as in the unsharded paths (which never read the .fai).
      for (size_t i = 0; ok && i < rows.size(); i++) {
        const FastaContig& r = rows[i];
        const uint64_t end = r.raw_offset + r.length + (r.length ? (r.length - 1) / r.line_bases : 0) * (uint64_t)(r.line_width - r.line_bases);
        char hb[2304];
        const uint64_t back = std::min<uint64_t>(r.raw_offset, sizeof hb);
        if (back < 2 || pread(fd, hb, back, (off_t)(r.raw_offset - back)) != (ssize_t)back || hb[back - 1] != '\n') { ok = false; break; }
        int64_t h = (int64_t)back - 2;   // start of the header line: the byte after the previous line break
        while (h >= 0 && hb[h] != '\n') h--;
        if (h < 0 && back < r.raw_offset) { ok = false; break; }   // a header longer than the buffer: not an index of this file
        const char* line = hb + h + 1;
        if (*line != '>') { ok = false; break; }
        std::string name(line + 1, (size_t)((hb + back - 1) - (line + 1)));
        if (!name.empty() && name.back() == '\r') {
          if (!crlf_as_lf) { ok = false; break; }   // carriage returns that stay in names and bases: the general parser's case
          name.pop_back();
        }
        name = name.substr(0, name.find_first_of(" \t"));
        if (abbr_of_chr(name) != keys[i]) { ok = false; break; }
        if (end < size) {
          char c = 0;
          if (pread(fd, &c, 1, (off_t)end) != 1 || (c != '\n' && c != '\r' && c != '>')) { ok = false; break; }
        }
        if (r.length) {   // the last base is a base
          char c = 0;
          if (pread(fd, &c, 1, (off_t)(end - 1)) != 1 || c == '\n' || c == '\r' || c == '>') { ok = false; break; }
        }
      }
      if (!ok) { keys.clear(); rows.clear(); }
    }
  }
  // 2. header scan on the host: '>' at a line start; '@' headers and ';' comment lines are left to the general parser
  if (!ok) {
    const int nt = std::max(1, threads);
    const uint64_t kChunk = 8u << 20;
    const uint64_t nchunks = (size + kChunk - 1) / kChunk;
    std::vector<std::vector<uint64_t>> found((size_t)nt);
    std::vector<char> odd((size_t)nt, 0);
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++)
      pool.emplace_back([&, t]() {
        std::vector<char> buf(kChunk + 1);
        for (uint64_t c = (uint64_t)t; c < nchunks; c += (uint64_t)nt) {
          // one byte of lead-in: the line-start test of the chunk's first byte
          const uint64_t a = c * kChunk, lead = a ? 1 : 0, n = std::min<uint64_t>(kChunk, size - a);
          uint64_t have = 0;
          while (have < n + lead) {
            const ssize_t got = pread(fd, buf.data() + have, n + lead - have, (off_t)(a - lead + have));
            if (got <= 0) { odd[(size_t)t] = 1; return; }
            have += (uint64_t)got;
          }
          const char* p0 = buf.data() + lead;
          for (const char* p = p0; p < p0 + n;) {
            const bool at_start = (p == p0) ? (a == 0 || p[-1] == '\n') : true;
            if (at_start) {
              if (*p == '>') found[(size_t)t].push_back(a + (uint64_t)(p - p0));
              else if (*p == '@' || *p == ';') odd[(size_t)t] = 1;
            }
            const char* nl = (const char*)memchr(p, '\n', (size_t)(p0 + n - p));
            if (!nl) break;
            p = nl + 1;
          }
        }
      });
    for (std::thread& th : pool) th.join();
    std::vector<uint64_t> hdr;
    bool plain = true;
    for (int t = 0; t < nt; t++) { plain &= !odd[(size_t)t]; hdr.insert(hdr.end(), found[(size_t)t].begin(), found[(size_t)t].end()); }
    std::sort(hdr.begin(), hdr.end());
    ok = plain && !hdr.empty() && rows_from_headers(fd, size, hdr, keys, rows, crlf_as_lf);
  }
  ::close(fd);
  if (ok) {   // a repeated name: ownership is by name, so such a file is ingested whole by every rank (open_on_device)
    std::vector<std::string> sorted_keys = keys;
    std::sort(sorted_keys.begin(), sorted_keys.end());
    if (std::adjacent_find(sorted_keys.begin(), sorted_keys.end()) != sorted_keys.end()) ok = false;
  }
  if (!ok) return false;
  on_device = true;
  names.clear(); seqs.clear(); contigs = rows; contig_of.clear(); dev_row.clear();
  for (size_t i = 0; i < keys.size(); i++) {
    names.push_back(keys[i]);
    contig_of[keys[i]] = (uint32_t)i;
  }
  return true;
}

void Fasta::open_owned_on_device(const std::string& ref_file, sg_ctx* ctx, int threads, const std::vector<char>& owned) {
  const std::string path = plain_path(ref_file);
  int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) throw Error("could not open " + path);
  // device image: the owned contigs' line ranges back to back (64-byte aligned starts)
  struct Piece { uint64_t file_off, bytes, dev_off; };
  std::vector<Piece> pieces;
  std::vector<sg_contig> tab;
  dev_row.assign(contigs.size(), -1);
  uint64_t total = 0;
  for (size_t i = 0; i < contigs.size(); i++) {
    if (i >= owned.size() || !owned[i]) continue;
    const FastaContig& r = contigs[i];
    const uint64_t bytes = r.length + (r.length ? (r.length - 1) / r.line_bases : 0) * (uint64_t)(r.line_width - r.line_bases);
    dev_row[i] = (int32_t)tab.size();
    tab.push_back(sg_contig{total, r.length, r.line_bases, r.line_width});
    pieces.push_back(Piece{r.raw_offset, bytes, total});
    total += (bytes + 63) & ~(uint64_t)63;
  }
  void* stage[2] = {nullptr, nullptr};
  try {
    const bool trace = getenv("SIMU_TRACE_LOAD") != nullptr;   // phase times of the ingest on stderr
    using Clk = std::chrono::steady_clock;
    auto secs = [](Clk::time_point a) { return std::chrono::duration<double>(Clk::now() - a).count(); };
    auto t_ph = Clk::now();
    eng_check(ctx, sg_reference_begin(ctx, total), "sg_reference_begin");
    const double t_begin = secs(t_ph);
    t_ph = Clk::now();
    const uint64_t kChunk = 64u << 20;
    for (int i = 0; i < 2 && total; i++) eng_check(ctx, sg_host_alloc(ctx, std::min<uint64_t>(kChunk, total), &stage[i]), "sg_host_alloc");
    const double t_pin = secs(t_ph);
    double t_read = 0, t_wait = 0;
    t_ph = Clk::now();
    int cur = 0;
    ReaderPool pool((int)std::min<uint64_t>((uint64_t)std::max(1, threads), 16));
    for (const Piece& pc : pieces)
      for (uint64_t off = 0; off < pc.bytes; off += kChunk, cur ^= 1) {
        const uint64_t n = std::min<uint64_t>(kChunk, pc.bytes - off);
        auto t1 = Clk::now();
        parallel_pread(pool, fd, (uint8_t*)stage[cur], pc.file_off + off, n);  // overlaps the copy of the other buffer
        t_read += secs(t1);
        t1 = Clk::now();
        eng_check(ctx, sg_sync(ctx), "sg_sync");
        t_wait += secs(t1);
        eng_check(ctx, sg_reference_chunk(ctx, pc.dev_off + off, stage[cur], n), "sg_reference_chunk");
      }
    eng_check(ctx, sg_sync(ctx), "sg_sync");
    const double t_stream = secs(t_ph);
    t_ph = Clk::now();
    eng_check(ctx, sg_reference_commit(ctx, tab.data(), (uint32_t)tab.size()), "sg_reference_commit");
    if (trace)
      fprintf(stderr, "load trace: device buffer %.3fs, pinned staging %.3fs, stream %.3fs (pread %.3fs, waiting for copies %.3fs) of %.2f GB, commit %.3fs\n",
              t_begin, t_pin, t_stream, t_read, t_wait, total / 1e9, secs(t_ph));
  } catch (...) {
    ::close(fd);
    for (void* b : stage) if (b) sg_host_free(ctx, b);
    throw;
  }
  ::close(fd);
  for (void* b : stage) if (b) sg_host_free(ctx, b);
  streamed = true;
}

}  // namespace simu
