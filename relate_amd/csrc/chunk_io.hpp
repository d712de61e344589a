// Readers and writers of the chunk files the painting stage consumes, in the
// formats Data::MakeChunks writes them (src/data.cpp:272-286,:496-526) and
// Data::Data / Paint read them (src/data.cpp:88-99,:541-550; src/data.hpp:78-88;
// pipeline/Paint.cpp:21-29).  Host-only code.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace relate {
namespace io {

// parameters_c<c>.bin: int32 N, int32 L, int32 W+1, int32 wb[W+1]
bool read_parameters(const std::string& path, int& N, int& L, std::vector<int32_t>& wb, std::string& err);
bool write_parameters(const std::string& path, int N, int L, const std::vector<int32_t>& wb, std::string& err);
// chunk_<c>.hap: uint64 L, uint64 N, char[L*N] ('0'/'1', row = SNP)   (CollapsedMatrix<char>::DumpToFile)
bool read_hap(const std::string& path, int& N, int& L, std::vector<uint8_t>& seq, std::string& err);
bool write_hap(const std::string& path, int N, int L, const uint8_t* seq, std::string& err);
// WriteVectorAsBin / ReadVectorFromBin: uint32 size, T[size]
bool read_vector(const std::string& path, std::vector<double>& v, std::string& err);
bool write_vector(const std::string& path, const std::vector<double>& v, std::string& err);
bool write_vector(const std::string& path, const std::vector<int32_t>& v, std::string& err);
// chunk_<c>.state: int32 L, int32[L]  (written by hand in MakeChunks, src/data.cpp:286,:320-358)
bool write_state(const std::string& path, const std::vector<int32_t>& v, std::string& err);

}  // namespace io
}  // namespace relate
