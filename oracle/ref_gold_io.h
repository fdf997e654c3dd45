// oracle/ref_gold_io.h — TEST INFRASTRUCTURE ONLY (golden generation).
//
// Container for the binary goldens the reference-compiled harnesses emit (ref_hot_harness.cpp,
// ref_rng_harness.cpp).  Our own format, nothing of the reference in it:
//
//   char     magic[8] = "PTGOLD01"
//   uint32   nsections
//   nsections x { char name[24]; uint32 rows; uint32 cols; }
//   then every section's rows*cols little-endian 32-bit words, in table order.
//
// Floats are stored as their bit patterns.  tests/golden_io.py reads it back with numpy.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <deque>
#include <vector>

namespace gold {

struct Section {
  std::string name;
  uint32_t rows = 0, cols = 0;
  std::vector<uint32_t> w;
};

inline uint32_t fbits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
inline float bitsf(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

struct File {
  std::deque<Section> sec;  // deque: add() hands out references that must survive later add()s
  Section& add(const char* name, uint32_t cols) {
    sec.emplace_back();
    sec.back().name = name;
    sec.back().cols = cols;
    return sec.back();
  }
  const Section* find(const char* name) const {
    for (const Section& s : sec)
      if (s.name == name) return &s;
    return nullptr;
  }
  bool write(const char* path) const {
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    fwrite("PTGOLD01", 1, 8, f);
    uint32_t n = (uint32_t)sec.size();
    fwrite(&n, 4, 1, f);
    for (const Section& s : sec) {
      char name[24] = {0};
      strncpy(name, s.name.c_str(), 23);
      fwrite(name, 1, 24, f);
      uint32_t rows = s.cols ? (uint32_t)(s.w.size() / s.cols) : 0;
      fwrite(&rows, 4, 1, f);
      fwrite(&s.cols, 4, 1, f);
    }
    for (const Section& s : sec)
      if (!s.w.empty()) fwrite(s.w.data(), 4, s.w.size(), f);
    return fclose(f) == 0;
  }
  bool read(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char magic[8];
    uint32_t n = 0;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "PTGOLD01", 8) != 0 || fread(&n, 4, 1, f) != 1) {
      fclose(f);
      return false;
    }
    sec.assign(n, Section());
    for (Section& s : sec) {
      char name[25] = {0};
      if (fread(name, 1, 24, f) != 24 || fread(&s.rows, 4, 1, f) != 1 || fread(&s.cols, 4, 1, f) != 1) {
        fclose(f);
        return false;
      }
      s.name = name;
    }
    for (Section& s : sec) {
      s.w.resize((size_t)s.rows * s.cols);
      if (!s.w.empty() && fread(s.w.data(), 4, s.w.size(), f) != s.w.size()) {
        fclose(f);
        return false;
      }
    }
    fclose(f);
    return true;
  }
};

}  // namespace gold
