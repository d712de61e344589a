// relate-gpu: command-line front end of the painting stage, with the options of
// `Relate --mode Paint` (pipeline/Relate.cpp:18-41,:75-82): -o/--output,
// --chunk_index, --painting theta,rho; plus --device (default: all devices).
// `--mode MakeChunks` takes the options of the reference's stage of that name
// (pipeline/MakeChunks.cpp:15-72: --haps, --sample, --map, -o, --memory, --transversion)
// and writes the same chunk files from bit rows; it needs no GPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include <sys/stat.h>

#include "../../include/relate_gpu.h"

static void usage()
{
  fprintf(stderr,
          "usage: relate-gpu --mode Paint -o <output dir> [--chunk_index <c>] [--painting <theta>,<rho>] [--device <d>]\n"
          "       relate-gpu --mode MakeChunks --haps <file> --sample <file> --map <file> -o <new output dir>\n"
          "                  [--memory <GB, default 5>] [--transversion]\n");
}

int main(int argc, char** argv)
{
  std::string mode, output, painting, haps, sample, map;
  int chunk = 0, device = -1;
  double memory = 5.0;  // Data::MakeChunks' default (src/data.hpp: min_memory = 5)
  bool transversion = false;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
    const char* v = nullptr;
    if (a == "--mode") { if (!(v = next())) break; mode = v; }
    else if (a == "-o" || a == "--output") { if (!(v = next())) break; output = v; }
    else if (a == "--chunk_index") { if (!(v = next())) break; chunk = atoi(v); }
    else if (a == "--painting") { if (!(v = next())) break; painting = v; }
    else if (a == "--device") { if (!(v = next())) break; device = atoi(v); }
    else if (a == "--haps") { if (!(v = next())) break; haps = v; }
    else if (a == "--sample") { if (!(v = next())) break; sample = v; }
    else if (a == "--map") { if (!(v = next())) break; map = v; }
    else if (a == "--memory") { if (!(v = next())) break; memory = atof(v); }
    else if (a == "--transversion") { transversion = true; }
    else if (a == "-h" || a == "--help") { usage(); return 0; }
    else { fprintf(stderr, "unknown option %s\n", a.c_str()); usage(); return 1; }
  }
  if (mode == "MakeChunks") {
    if (haps.empty() || sample.empty() || map.empty() || output.empty()) {
      fprintf(stderr, "Not enough arguments supplied.\nNeeded: haps, sample, map, output. Optional: memory, transversion.\n");
      usage();
      return 1;
    }
    struct stat sb;
    if (stat(output.c_str(), &sb) == 0) {  // pipeline/MakeChunks.cpp:38-42
      fprintf(stderr, "Error: Directory %s already exists. Relate will use this directory to store temporary files.\n", output.c_str());
      return 1;
    }
    relate_haps* h = nullptr;
    int rc = relate_haps_open(haps.c_str(), sample.c_str(), map.c_str(), memory, &h);
    if (rc == RELATE_OK) {
      if (mkdir(output.c_str(), 0777) != 0) {
        fprintf(stderr, "relate-gpu: cannot create directory %s\n", output.c_str());
        relate_haps_close(h);
        return 1;
      }
      rc = relate_haps_write_chunks(h, output.c_str(), transversion ? 0 : 1);
    }
    if (rc != RELATE_OK) {
      fprintf(stderr, "relate-gpu: %s (code %d)\n", relate_last_error(), rc);
      relate_haps_close(h);
      return 1;
    }
    int N = 0, L = 0, chunks = 0;
    relate_haps_dims(h, &N, &L, &chunks);
    fprintf(stderr, "%d haplotypes, %d SNPs, %d chunk(s) written to %s\n", N, L, chunks, output.c_str());
    relate_haps_close(h);
    return 0;
  }
  if (mode != "Paint" || output.empty()) {
    usage();
    return 1;
  }
  double theta = 0, rho = 0;
  const double *pt = nullptr, *pr = nullptr;
  if (!painting.empty()) {
    // pipeline/Paint.cpp:35-56 parses both numbers with std::stof, i.e. through float
    const size_t comma = painting.find(',');
    if (comma == std::string::npos) {
      fprintf(stderr, "--painting expects theta,rho\n");
      return 1;
    }
    try {
      theta = (double)std::stof(painting.substr(0, comma));
      rho = (double)std::stof(painting.substr(comma + 1));
    } catch (const std::exception&) {  // std::invalid_argument / std::out_of_range
      fprintf(stderr, "--painting expects two numbers, theta,rho (got \"%s\")\n", painting.c_str());
      return 1;
    }
    pt = &theta;
    pr = &rho;
  }
  const int rc = relate_paint_chunk(output.c_str(), chunk, pt, pr, device);
  if (rc != RELATE_OK) {
    fprintf(stderr, "relate-gpu: %s (code %d)\n", relate_last_error(), rc);
    return 1;
  }
  return 0;
}
