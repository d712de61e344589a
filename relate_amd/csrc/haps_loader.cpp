// Loader of phased haplotypes: .haps / .sample / genetic map -> bit rows, recombination distances, chunks and
// windows, without the one-byte-per-allele matrices Data::MakeChunks keeps and writes (src/data.cpp:134-528).
// The rules restated here, each beside its reference lines:
//   N from the .sample file, L from the .haps file          src/data.hpp:115-149
//   one SNP per line, alleles '0' / '1' in order            src/data.cpp:553-580
//   chunk and window cuts (memory rule, 500 windows,
//   overlap of 20 000 SNPs between chunks)                  src/data.cpp:145-267,:288-318
//   dist from consecutive positions                         src/data.cpp:397-415
//   the genetic map and its interpolation (rpos), r         src/data.cpp:452-491,:598-632
//   the files of every chunk                                src/data.cpp:270-368,:380-391,:437-449,:496-526
// Host-only code; the device side is relate_plan_load_haps_chunk (relate_gpu.cpp).
#include "haps_loader.hpp"

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "../../include/relate_gpu.h"
#include "chunk_io.hpp"

namespace relate {
int row_stride_words(int N);  // paint_kernels.hip: words per packed row
}

namespace {

using relate::set_last_error;

// A text file that may be gzip-compressed: as gzip::open (src/data.cpp:8-69), compressed files are read
// through `gunzip -c`.
struct TextFile {
  FILE* fp = nullptr;
  bool piped = false;
  char* line = nullptr;
  size_t cap = 0;
  ~TextFile() { close(); }
  void close()
  {
    if (fp) piped ? (void)pclose(fp) : (void)fclose(fp);
    fp = nullptr;
    free(line);
    line = nullptr;
  }
  bool open(const std::string& path)
  {
    FILE* probe = fopen(path.c_str(), "rb");
    if (!probe) return false;
    unsigned char magic[3] = {0, 0, 0};
    const size_t got = fread(magic, 1, 3, probe);
    fclose(probe);
    if (got == 3 && magic[0] == 0x1f && magic[1] == 0x8b && magic[2] == 0x08) {
      std::string quoted = "'";
      for (char c : path) quoted += (c == '\'') ? std::string("'\\''") : std::string(1, c);
      quoted += "'";
      fp = popen(("gunzip -c " + quoted).c_str(), "r");
      piped = true;
    } else {
      fp = fopen(path.c_str(), "r");
      piped = false;
    }
    return fp != nullptr;
  }
  // next line without its '\n'; *terminated says whether the newline was there
  bool next(char** out, size_t* len, bool* terminated)
  {
    const ssize_t n = getline(&line, &cap, fp);
    if (n < 0) return false;
    *terminated = n > 0 && line[n - 1] == '\n';
    *len = (size_t)n - (*terminated ? 1 : 0);
    line[*len] = '\0';
    *out = line;
    return true;
  }
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

// next whitespace-delimited token of [*p, end); false at the end
bool token(char** p, char* end, char** tok, size_t* len)
{
  char* q = *p;
  while (q < end && is_space(*q)) q++;
  if (q >= end) return false;
  char* t = q;
  while (q < end && !is_space(*q)) q++;
  *tok = t;
  *len = (size_t)(q - t);
  *p = q;
  return true;
}

// Whole file as whitespace-delimited tokens (what a chain of fscanf("%s") sees) and its number of newlines.
bool read_tokens(const std::string& path, std::vector<std::string>& toks, long long* newlines)
{
  TextFile f;
  if (!f.open(path)) return false;
  char* s;
  size_t len;
  bool term;
  *newlines = 0;
  while (f.next(&s, &len, &term)) {
    if (term) (*newlines)++;
    char *p = s, *t;
    size_t tl;
    while (token(&p, s + len, &t, &tl)) toks.emplace_back(t, tl);
  }
  return true;
}

// N from the sample file (src/data.hpp:118-131): two header triples, then one triple per individual;
// ID_1 == ID_2 marks a diploid (two haplotypes), anything else one haplotype.
int count_haplotypes(const std::string& path, int* N)
{
  std::vector<std::string> t;
  long long nl;
  if (!read_tokens(path, t, &nl)) return set_last_error(RELATE_EIO, "cannot open " + path);
  if (t.size() < 6) return set_last_error(RELATE_EINVAL, path + ": expected two header lines of three fields");
  long long n = 0;
  for (size_t i = 6; i + 2 < t.size(); i += 3) n += (t[i] == t[i + 1]) ? 2 : 1;
  if (n < 2 || n > (1ll << 30)) return set_last_error(RELATE_EINVAL, path + ": fewer than two haplotypes");
  *N = (int)n;
  return RELATE_OK;
}

// The genetic map (src/data.cpp:598-632): a header of three fields, then "position rate cM" per line; the position
// is read as a double and truncated.
int read_map(const std::string& path, std::vector<int>& bp, std::vector<double>& gen_pos)
{
  std::vector<std::string> t;
  long long nl;
  if (!read_tokens(path, t, &nl)) return set_last_error(RELATE_EIO, "cannot open " + path);
  const long long lines = nl - 1;  // the header does not count
  if (lines < 2) return set_last_error(RELATE_EINVAL, path + ": a map needs at least two positions");
  if ((long long)t.size() < 3 + 3 * lines) return set_last_error(RELATE_EINVAL, path + ": fewer than three fields on a line");
  bp.resize((size_t)lines);
  gen_pos.resize((size_t)lines);
  for (long long i = 0; i < lines; i++) {
    char *e0, *e2;
    const std::string &a = t[3 + 3 * i], &c = t[3 + 3 * i + 2];
    const double fbp = strtod(a.c_str(), &e0);
    const double g = strtod(c.c_str(), &e2);
    if (e0 == a.c_str() || e2 == c.c_str()) return set_last_error(RELATE_EINVAL, path + ": not a number on line " + std::to_string(i + 2));
    bp[(size_t)i] = (int)fbp;
    gen_pos[(size_t)i] = g;
    if (i > 0 && bp[(size_t)i] < bp[(size_t)i - 1])
      return set_last_error(RELATE_EINVAL, path + ": positions decrease at line " + std::to_string(i + 2));  // the reference asserts (:469)
  }
  return RELATE_OK;
}

// rpos (src/data.cpp:452-479): the genetic position of every SNP and of one base past the last, in Morgans,
// linear between the two map entries around it (the last interval also serves everything beyond the map).
void interpolate_map(const std::vector<int>& mbp, const std::vector<double>& gen, const std::vector<int32_t>& bp,
                     std::vector<double>& rpos)
{
  const size_t n = bp.size();
  rpos.resize(n);
  size_t i = 0, map_pos = 0;
  if (mbp[0] > bp[0]) {
    rpos[0] = gen[0] * 1e-2;
    i = 1;
  }
  for (; i < n; i++) {
    while (mbp[map_pos + 1] <= bp[i] && map_pos < mbp.size() - 2) map_pos++;
    if (mbp[map_pos + 1] - mbp[map_pos] == 0 || mbp[map_pos] > bp[i]) {
      rpos[i] = gen[map_pos] * 1e-2;
    } else {
      rpos[i] = ((bp[i] - mbp[map_pos]) / ((double)(mbp[map_pos + 1] - mbp[map_pos])) * (gen[map_pos + 1] - gen[map_pos]) +
                 gen[map_pos]) *
                1e-2;
    }
  }
}

// The cuts of MakeChunks (src/data.cpp:145-267): a window closes at the SNP at which the sum of derived * (N + 1)
// over its SNPs has reached the budget (and it holds more than 10 SNPs); a chunk closes at 500 windows (those it
// shares with the previous chunk included), at max_chunk_size SNPs or at the end; the next chunk starts 20 000
// SNPs back and inherits the window boundaries inside that overlap.
int cut_chunks(relate_haps* h, double memory_gb)
{
  const int N = h->N, L = h->L;
  const float min_memory = (float)memory_gb;  // the reference's parameter is a float
  const double budget = (min_memory)*1e9 / 4.0 - (2.0 * N * N + 3.0 * N);
  if (budget <= 0) return set_last_error(RELATE_ERANGE, "need a larger memory allowance for this number of haplotypes");
  const int windows_per_section = 500, overlap = 20000;
  int max_chunk_size = (int)std::min<double>((double)L + 1.0, budget / N);
  if (min_memory >= 100) max_chunk_size = 2500000;
  if (max_chunk_size < 1) return set_last_error(RELATE_ERANGE, "need a larger memory allowance for this number of haplotypes");

  const int full = N >> 6, rem = N & 63;
  auto derived = [&](int s) {
    const uint64_t* row = h->rows.data() + (size_t)s * h->row_stride;
    int c = 0;
    for (int w = 0; w < full; w++) c += __builtin_popcountll(row[w]);
    if (rem) c += __builtin_popcountll(row[full] & ((1ull << rem) - 1ull));
    return c;
  };

  std::vector<int> wbnd(windows_per_section + 1), wover;
  std::vector<int> section_start{0};
  int snp = 0, num_windows = 0, num_over = 0;
  double actual_min = 0.0;
  while (snp < L) {
    wover.clear();
    num_over = 0;
    if (snp > 0) {
      if (snp - section_start.back() < overlap)
        return set_last_error(RELATE_ERANGE, "a chunk closed before it held the 20000 SNPs the next one overlaps: larger memory allowance needed");
      const int begin = snp - overlap;
      section_start.push_back(begin);
      wover.push_back(begin);
      for (int i = 0; i < num_windows; i++)
        if (wbnd[i] > begin) wover.push_back(wbnd[i]);
      num_over = (int)wover.size();
      if (num_over >= windows_per_section - 1)
        return set_last_error(RELATE_ERANGE, "too many windows inside the overlap of two chunks: larger memory allowance needed");
    }
    const int snp_begin = snp;
    double window_memory = 0.0;
    int chunk_size = 0, in_window = 0;
    wbnd[0] = snp_begin;
    num_windows = 1;
    while (num_windows + num_over < windows_per_section && chunk_size < max_chunk_size && snp < L) {
      window_memory += (double)derived(snp) * (N + 1);  // the reference multiplies two ints: the same value wherever that is defined
      if (window_memory >= budget && in_window > 10) {
        actual_min = std::max(actual_min, window_memory);
        in_window = 0;
        window_memory = 0.0;
        wbnd[num_windows++] = snp;
      }
      snp++;
      in_window++;
      chunk_size++;
    }
    actual_min = std::max(actual_min, window_memory);
    wbnd[num_windows] = snp;

    relate::HapsChunk c;
    if (snp_begin == 0) {
      c.first_snp = 0;
      c.L = chunk_size;
      c.wb.assign(wbnd.begin(), wbnd.begin() + num_windows + 1);
    } else {
      c.first_snp = wover[0];
      c.L = chunk_size + overlap;
      for (int v : wover) c.wb.push_back(v - wover[0]);
      for (int i = 0; i <= num_windows; i++) c.wb.push_back(wbnd[i] - wover[0]);
    }
    h->chunks.push_back(std::move(c));
  }
  h->actual_min_memory_gb = (actual_min + (2.0 * N * N + 3.0 * N)) * 4.0 / 1e9;
  return RELATE_OK;
}

int parse_haps(relate_haps* h, const std::string& path)
{
  TextFile f;
  if (!f.open(path)) return set_last_error(RELATE_EIO, "cannot open " + path);
  const int N = h->N, stride = h->row_stride;
  const int full = N >> 6, rem = N & 63;
  char* s;
  size_t len;
  bool term;
  long long lineno = 0;
  while (f.next(&s, &len, &term)) {
    lineno++;
    if (!term) break;  // L is the number of newlines (src/data.hpp:136-141): an unterminated last line is not a SNP
    char *p = s, *end = s + len, *t[5];
    size_t tl[5];
    for (int i = 0; i < 5; i++)
      if (!token(&p, end, &t[i], &tl[i]))
        return set_last_error(RELATE_EINVAL, path + ": line " + std::to_string(lineno) + " has fewer than five leading fields");
    char* e;
    errno = 0;
    const long pos = strtol(t[2], &e, 10);
    if (e == t[2] || errno || pos < 0 || pos > 0x7fffffffl)
      return set_last_error(RELATE_EINVAL, path + ": line " + std::to_string(lineno) + ": position is not an integer");
    if (h->L == 0x7fffffff) return set_last_error(RELATE_ERANGE, path + ": too many SNPs");
    const size_t base = h->rows.size();
    h->rows.resize(base + (size_t)stride, 0);
    uint64_t* row = h->rows.data() + base;
    int n = 0;
    while (p < end && n < N) {
      // four alleles at a time while the text is the usual " 0 1 1 0": eight bytes, spaces at the even ones and
      // '0' / '1' at the odd ones, whose low bits are the alleles
      while ((n & 3) == 0 && n + 4 <= N && p + 8 <= end) {
        uint64_t x;
        memcpy(&x, p, 8);
        if ((x & 0xFEFFFEFFFEFFFEFFull) != 0x3020302030203020ull) break;
        const uint64_t b = (x >> 8) & 0x0001000100010001ull;
        const uint64_t four = (b | (b >> 15) | (b >> 30) | (b >> 45)) & 0xFull;
        row[n >> 6] |= four << (n & 63);
        n += 4;
        p += 8;
      }
      if (p >= end || n >= N) break;
      // anything else character by character, as haps::ReadSNP does: '0' and '1' count, the rest is skipped
      if (*p == '1') {
        row[n >> 6] |= 1ull << (n & 63);
        n++;
      } else if (*p == '0') {
        n++;
      }
      p++;
    }
    if (n != N)
      return set_last_error(RELATE_EINVAL, path + ": line " + std::to_string(lineno) + " holds " + std::to_string(n) + " alleles, the sample file announces " + std::to_string(N));
    // padding bits are set: states that never mismatch (what the device-side packing writes too)
    if (rem) row[full] |= ~((1ull << rem) - 1ull);
    for (int w = full + (rem ? 1 : 0); w < stride; w++) row[w] = ~0ull;
    h->bp.push_back((int32_t)pos);
    h->rsid.emplace_back(t[1], tl[1]);
    h->ancestral.emplace_back(t[3], tl[3]);
    h->alternative.emplace_back(t[4], tl[4]);
    h->L++;
  }
  if (h->L < 2) return set_last_error(RELATE_EINVAL, path + ": fewer than two SNPs");
  return RELATE_OK;
}

bool is_transition(const std::string& a, const std::string& b)
{
  return (a == "C" && b == "T") || (a == "T" && b == "C") || (a == "G" && b == "A") || (a == "A" && b == "G");
}

}  // namespace

extern "C" {

int relate_haps_open(const char* haps_path, const char* sample_path, const char* map_path, double memory_gb,
                     relate_haps** out)
{
  if (!haps_path || !sample_path || !map_path || !out) return set_last_error(RELATE_EINVAL, "null argument");
  *out = nullptr;
  std::unique_ptr<relate_haps> h(new relate_haps());
  try {
    if (int rc = count_haplotypes(sample_path, &h->N)) return rc;
    h->row_stride = relate::row_stride_words(h->N);
    if (int rc = parse_haps(h.get(), haps_path)) return rc;
    const int L = h->L;
    // positions: one more entry, a base past the last SNP (src/data.cpp:370)
    h->bp.push_back(h->bp[(size_t)L - 1] + 1);
    h->dist.resize((size_t)L);
    for (int s = 0; s + 1 < L; s++) {
      h->dist[(size_t)s] = h->bp[(size_t)s + 1] - h->bp[(size_t)s];
      if (h->dist[(size_t)s] <= 0)
        return set_last_error(RELATE_EINVAL, "SNPs are not sorted by position, or two share a position, at BP " + std::to_string(h->bp[(size_t)s]));
    }
    h->dist[(size_t)L - 1] = 1;
    if (int rc = cut_chunks(h.get(), memory_gb)) return rc;
    std::vector<int> mbp;
    std::vector<double> gen;
    if (int rc = read_map(map_path, mbp, gen)) return rc;
    interpolate_map(mbp, gen, h->bp, h->rpos);
    h->r.resize((size_t)L);
    const double lower_bound = 1e-10;
    for (int s = 0; s < L; s++) {
      double v = h->rpos[(size_t)s + 1] - h->rpos[(size_t)s];
      if (v < lower_bound) v = lower_bound;
      h->r[(size_t)s] = v * 2500;
    }
  } catch (const std::bad_alloc&) {
    return set_last_error(RELATE_ENOMEM, "out of host memory while reading the data set");
  }
  *out = h.release();
  return RELATE_OK;
}

void relate_haps_close(relate_haps* h) { delete h; }

int relate_haps_dims(const relate_haps* h, int* N, int* L, int* num_chunks)
{
  if (!h) return set_last_error(RELATE_EINVAL, "null data set");
  if (N) *N = h->N;
  if (L) *L = h->L;
  if (num_chunks) *num_chunks = (int)h->chunks.size();
  return RELATE_OK;
}

int relate_haps_chunk_dims(const relate_haps* h, int chunk, int* first_snp, int* L_chunk, int* num_windows)
{
  if (!h || chunk < 0 || chunk >= (int)h->chunks.size()) return set_last_error(RELATE_EINVAL, "no such chunk");
  const relate::HapsChunk& c = h->chunks[(size_t)chunk];
  if (first_snp) *first_snp = c.first_snp;
  if (L_chunk) *L_chunk = c.L;
  if (num_windows) *num_windows = (int)c.wb.size() - 1;
  return RELATE_OK;
}

int relate_haps_chunk_windows(const relate_haps* h, int chunk, int32_t* wb)
{
  if (!h || !wb || chunk < 0 || chunk >= (int)h->chunks.size()) return set_last_error(RELATE_EINVAL, "no such chunk");
  const relate::HapsChunk& c = h->chunks[(size_t)chunk];
  std::copy(c.wb.begin(), c.wb.end(), wb);
  return RELATE_OK;
}

int relate_haps_positions(const relate_haps* h, int32_t* bp, int32_t* dist, double* rpos, double* r)
{
  if (!h) return set_last_error(RELATE_EINVAL, "null data set");
  if (bp) std::copy(h->bp.begin(), h->bp.end(), bp);
  if (dist) std::copy(h->dist.begin(), h->dist.end(), dist);
  if (rpos) std::copy(h->rpos.begin(), h->rpos.end(), rpos);
  if (r) std::copy(h->r.begin(), h->r.end(), r);
  return RELATE_OK;
}

int relate_haps_rows(const relate_haps* h, const uint64_t** rows, int* row_stride_words)
{
  if (!h || !rows || !row_stride_words) return set_last_error(RELATE_EINVAL, "null argument");
  *rows = h->rows.data();
  *row_stride_words = h->row_stride;
  return RELATE_OK;
}

int relate_haps_write_chunks(const relate_haps* h, const char* output_dir, int use_transitions)
{
  if (!h || !output_dir) return set_last_error(RELATE_EINVAL, "null argument");
  using namespace relate::io;
  const std::string dir(output_dir);
  const int N = h->N, L = h->L;
  std::string err;
  try {
    // the mutation class of every SNP (src/data.cpp:323-334,:346-357): 0 for transitions unless they are to be used
    std::vector<int32_t> state((size_t)L, 1);
    if (!use_transitions)
      for (int s = 0; s < L; s++) state[(size_t)s] = is_transition(h->ancestral[(size_t)s], h->alternative[(size_t)s]) ? 0 : 1;
    std::vector<uint8_t> seq;
    for (size_t c = 0; c < h->chunks.size(); c++) {
      const relate::HapsChunk& ch = h->chunks[c];
      const std::string base = dir + "/chunk_" + std::to_string(c);
      const size_t a = (size_t)ch.first_snp, b = a + (size_t)ch.L;
      seq.assign((size_t)ch.L * N, '0');
      for (size_t s = a; s < b; s++) {
        const uint64_t* row = h->rows.data() + s * h->row_stride;
        uint8_t* o = seq.data() + (s - a) * N;
        for (int n = 0; n < N; n++) o[n] = (uint8_t)('0' + ((row[n >> 6] >> (n & 63)) & 1ull));
      }
      const std::vector<int32_t> bp(h->bp.begin() + a, h->bp.begin() + b), dist(h->dist.begin() + a, h->dist.begin() + b),
          st(state.begin() + a, state.begin() + b);
      const std::vector<double> r(h->r.begin() + a, h->r.begin() + b), rpos(h->rpos.begin() + a, h->rpos.begin() + b + 1);
      const bool ok = write_parameters(dir + "/parameters_c" + std::to_string(c) + ".bin", N, ch.L, ch.wb, err) &&
                      write_hap(base + ".hap", N, ch.L, seq.data(), err) && write_vector(base + ".bp", bp, err) &&
                      write_vector(base + ".dist", dist, err) && write_vector(base + ".r", r, err) &&
                      write_vector(base + ".rpos", rpos, err) && write_state(base + ".state", st, err);
      if (!ok) return set_last_error(RELATE_EIO, err);
    }
    // parameters.bin (src/data.cpp:380-391): N, L, number of chunks, memory actually needed, first and last SNP of every chunk
    {
      std::unique_ptr<FILE, int (*)(FILE*)> f(fopen((dir + "/parameters.bin").c_str(), "wb"), fclose);
      if (!f) return set_last_error(RELATE_EIO, "cannot open " + dir + "/parameters.bin");
      const int32_t nc = (int32_t)h->chunks.size();
      std::vector<int32_t> first, last;
      for (const auto& ch : h->chunks) {
        first.push_back(ch.first_snp);
        last.push_back(ch.first_snp + ch.L);
      }
      const int32_t n = N, l = L;
      bool ok = fwrite(&n, 4, 1, f.get()) == 1 && fwrite(&l, 4, 1, f.get()) == 1 && fwrite(&nc, 4, 1, f.get()) == 1 &&
                fwrite(&h->actual_min_memory_gb, 8, 1, f.get()) == 1 && fwrite(first.data(), 4, first.size(), f.get()) == first.size() &&
                fwrite(last.data(), 4, last.size(), f.get()) == last.size();
      if (!ok) return set_last_error(RELATE_EIO, "short write to " + dir + "/parameters.bin");
    }
    // props.bin (src/data.cpp:437-449): per SNP its index, position, distance and three 1024-byte text fields
    // (the reference writes whatever follows the strings in memory; here the fields are zero-padded)
    {
      std::unique_ptr<FILE, int (*)(FILE*)> f(fopen((dir + "/props.bin").c_str(), "wb"), fclose);
      if (!f) return set_last_error(RELATE_EIO, "cannot open " + dir + "/props.bin");
      char field[1024];
      auto put = [&](const std::string& s) {
        memset(field, 0, sizeof field);
        memcpy(field, s.data(), std::min<size_t>(s.size(), sizeof field - 1));
        return fwrite(field, 1, sizeof field, f.get()) == sizeof field;
      };
      for (int32_t s = 0; s < L; s++) {
        const bool ok = fwrite(&s, 4, 1, f.get()) == 1 && fwrite(&h->bp[(size_t)s], 4, 1, f.get()) == 1 &&
                        fwrite(&h->dist[(size_t)s], 4, 1, f.get()) == 1 && put(h->rsid[(size_t)s]) &&
                        put(h->ancestral[(size_t)s]) && put(h->alternative[(size_t)s]);
        if (!ok) return set_last_error(RELATE_EIO, "short write to " + dir + "/props.bin");
      }
    }
  } catch (const std::bad_alloc&) {
    return set_last_error(RELATE_ENOMEM, "out of host memory while writing the chunk files");
  }
  return RELATE_OK;
}

}  // extern "C"
