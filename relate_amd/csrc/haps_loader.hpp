// A phased data set (.haps / .sample / genetic map) held as bit rows, cut into the chunks and windows that
// Data::MakeChunks would write (src/data.cpp:134-528).  Host-only code.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace relate {

// Sets the text relate_last_error() returns on the calling thread and passes `code` through.
int set_last_error(int code, const std::string& msg);

struct HapsChunk {
  int first_snp = 0;        // section_boundary_start[c]
  int L = 0;                // SNPs of the chunk, the overlap with the previous chunk included
  std::vector<int32_t> wb;  // window boundaries relative to first_snp, as parameters_c<c>.bin holds them
};

}  // namespace relate

struct relate_haps {
  int N = 0, L = 0;
  int row_stride = 0;          // 64-bit words per bit row (the plan's layout: bits past N are set)
  std::vector<uint64_t> rows;  // L * row_stride words, bit n of row s = haplotype n carries the derived allele at SNP s
  std::vector<int32_t> bp;     // L + 1 entries; the last is bp[L-1] + 1 (src/data.cpp:370)
  std::vector<int32_t> dist;   // L
  std::vector<double> rpos;    // L + 1
  std::vector<double> r;       // L
  std::vector<std::string> rsid, ancestral, alternative;
  std::vector<relate::HapsChunk> chunks;
  double actual_min_memory_gb = 0.0;  // what parameters.bin records (src/data.cpp:382-388)
};
