// relate-gpu: command-line front end of the painting stage, with the options of
// `Relate --mode Paint` (pipeline/Relate.cpp:18-41,:75-82): -o/--output,
// --chunk_index, --painting theta,rho; plus --device (default: all devices).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/relate_gpu.h"

static void usage()
{
  fprintf(stderr,
          "usage: relate-gpu --mode Paint -o <output dir> [--chunk_index <c>] [--painting <theta>,<rho>] [--device <d>]\n");
}

int main(int argc, char** argv)
{
  std::string mode, output, painting;
  int chunk = 0, device = -1;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
    const char* v = nullptr;
    if (a == "--mode") { if (!(v = next())) break; mode = v; }
    else if (a == "-o" || a == "--output") { if (!(v = next())) break; output = v; }
    else if (a == "--chunk_index") { if (!(v = next())) break; chunk = atoi(v); }
    else if (a == "--painting") { if (!(v = next())) break; painting = v; }
    else if (a == "--device") { if (!(v = next())) break; device = atoi(v); }
    else if (a == "-h" || a == "--help") { usage(); return 0; }
    else { fprintf(stderr, "unknown option %s\n", a.c_str()); usage(); return 1; }
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
