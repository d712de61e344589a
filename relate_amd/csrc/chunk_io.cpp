#include "chunk_io.hpp"

#include <cstdio>
#include <memory>

namespace relate {
namespace io {

namespace {
struct Closer {
  void operator()(FILE* f) const
  {
    if (f) fclose(f);
  }
};
using File = std::unique_ptr<FILE, Closer>;

File open_file(const std::string& path, const char* mode, std::string& err)
{
  File f(fopen(path.c_str(), mode));
  if (!f) err = "cannot open " + path;
  return f;
}
template <class T>
bool rd(FILE* f, T* p, size_t n)
{
  return fread(p, sizeof(T), n, f) == n;
}
template <class T>
bool wr(FILE* f, const T* p, size_t n)
{
  return fwrite(p, sizeof(T), n, f) == n;
}
}  // namespace

bool read_parameters(const std::string& path, int& N, int& L, std::vector<int32_t>& wb, std::string& err)
{
  File f = open_file(path, "rb", err);
  if (!f) return false;
  int32_t n = 0, l = 0, nw = 0;
  if (!rd(f.get(), &n, 1) || !rd(f.get(), &l, 1) || !rd(f.get(), &nw, 1) || nw < 2 || nw > (1 << 20)) {
    err = "malformed " + path;
    return false;
  }
  wb.resize(nw);
  if (!rd(f.get(), wb.data(), (size_t)nw)) {
    err = "truncated " + path;
    return false;
  }
  N = n;
  L = l;
  return true;
}

bool write_parameters(const std::string& path, int N, int L, const std::vector<int32_t>& wb, std::string& err)
{
  File f = open_file(path, "wb", err);
  if (!f) return false;
  const int32_t n = N, l = L, nw = (int32_t)wb.size();
  if (!wr(f.get(), &n, 1) || !wr(f.get(), &l, 1) || !wr(f.get(), &nw, 1) || !wr(f.get(), wb.data(), wb.size())) {
    err = "short write to " + path;
    return false;
  }
  return true;
}

bool read_hap(const std::string& path, int& N, int& L, std::vector<uint8_t>& seq, std::string& err)
{
  File f = open_file(path, "rb", err);
  if (!f) return false;
  uint64_t l = 0, n = 0;
  if (!rd(f.get(), &l, 1) || !rd(f.get(), &n, 1) || l < 2 || n < 2 || l > (1ull << 31) - 1 || n > (1ull << 31) - 1) {
    err = "malformed " + path;
    return false;
  }
  seq.resize((size_t)l * n);
  if (!rd(f.get(), seq.data(), seq.size())) {
    err = "truncated " + path;
    return false;
  }
  N = (int)n;
  L = (int)l;
  return true;
}

bool write_hap(const std::string& path, int N, int L, const uint8_t* seq, std::string& err)
{
  File f = open_file(path, "wb", err);
  if (!f) return false;
  const uint64_t l = (uint64_t)L, n = (uint64_t)N;
  if (!wr(f.get(), &l, 1) || !wr(f.get(), &n, 1) || !wr(f.get(), seq, (size_t)L * N)) {
    err = "short write to " + path;
    return false;
  }
  return true;
}

bool read_vector(const std::string& path, std::vector<double>& v, std::string& err)
{
  File f = open_file(path, "rb", err);
  if (!f) return false;
  uint32_t n = 0;
  if (!rd(f.get(), &n, 1)) {
    err = "malformed " + path;
    return false;
  }
  v.resize(n);
  if (!rd(f.get(), v.data(), (size_t)n)) {
    err = "truncated " + path;
    return false;
  }
  return true;
}

template <class T>
static bool write_vec_t(const std::string& path, const std::vector<T>& v, std::string& err)
{
  File f = open_file(path, "wb", err);
  if (!f) return false;
  const uint32_t n = (uint32_t)v.size();
  if (!wr(f.get(), &n, 1) || !wr(f.get(), v.data(), v.size())) {
    err = "short write to " + path;
    return false;
  }
  return true;
}
bool write_vector(const std::string& path, const std::vector<double>& v, std::string& err) { return write_vec_t(path, v, err); }
bool write_vector(const std::string& path, const std::vector<int32_t>& v, std::string& err) { return write_vec_t(path, v, err); }

bool write_state(const std::string& path, const std::vector<int32_t>& v, std::string& err)
{
  File f = open_file(path, "wb", err);
  if (!f) return false;
  const int32_t n = (int32_t)v.size();
  if (!wr(f.get(), &n, 1) || !wr(f.get(), v.data(), v.size())) {
    err = "short write to " + path;
    return false;
  }
  return true;
}

}  // namespace io
}  // namespace relate
