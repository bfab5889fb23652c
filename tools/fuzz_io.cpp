// AddressSanitizer / UBSan harness for csrc/io_native.cpp (CPU build only):
//   g++ -O1 -g -fsanitize=address,undefined -std=c++17 tools/fuzz_io.cpp -o /tmp/fuzz_io
//   /tmp/fuzz_io seed.jpg seed.example seed.tfrecord 20000
// Mutates the three seed inputs and feeds them to the JPEG decoder, the tf.Example parser and
// the TFRecord reader; any out-of-bounds access aborts with a sanitizer report.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../cap2det_amd/csrc/io_native.cpp"

static std::vector<uint8_t> slurp(const char* p) {
  std::vector<uint8_t> v; FILE* f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
  int c; while ((c = fgetc(f)) != EOF) v.push_back((uint8_t)c); fclose(f); return v;
}
static uint64_t rs = 88172645463325252ull;
static uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static std::vector<uint8_t> mutate(const std::vector<uint8_t>& in) {
  std::vector<uint8_t> d = in;
  switch (rnd() % 5) {
    case 0: for (int i = 0, n = 1 + rnd() % 8; i < n; ++i) d[rnd() % d.size()] = (uint8_t)rnd(); break;
    case 1: d.resize(1 + rnd() % d.size()); break;
    case 2: { size_t p = rnd() % d.size(); for (int i = 0; i < 4 && p + i < d.size(); ++i) d[p + i] = (uint8_t)rnd(); } break;
    case 3: { size_t p = rnd() % d.size(); d[p] = 0xff; if (p + 1 < d.size()) { const uint8_t m[] = {0xc0, 0xc4, 0xda, 0xdb, 0xdd, 0xd9, 0xd0, 0x00, 0xc2}; d[p + 1] = m[rnd() % 9]; } } break;
    default: { size_t p = rnd() % d.size(); d[p] = (uint8_t)(d[p] + 1); if (p + 2 < d.size()) d[p + 2] = 0xff; } break;
  }
  return d;
}
int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const std::vector<uint8_t> jpg = slurp(argv[1]), ex = slurp(argv[2]), rec = slurp(argv[3]);
  const long iters = atol(argv[4]);
  long okj = 0, oke = 0, okr = 0;
  const char* keys[] = {"image/source_id", "image/encoded", "image/proposal/bbox/ymin", "image/object/class/label", "image/caption/string", "x"};
  for (long it = 0; it < iters; ++it) {
    { std::vector<uint8_t> d = mutate(jpg); int h = 0, w = 0, c = 0;
      if (c2d_jpeg_info(d.data(), (long long)d.size(), &h, &w, &c) == 0 && h > 0 && w > 0 && (long long)h * w < (1 << 24)) {
        std::vector<uint8_t> out((size_t)h * w * 3), ws((size_t)c2d_jpeg_workspace_bytes(h, w));
        if (c2d_jpeg_decode_rgb(d.data(), (long long)d.size(), out.data(), h, w, ws.data(), (long long)ws.size()) == 0) ++okj;
      } }
    { std::vector<uint8_t> d = mutate(ex); int kinds[6]; long long counts[6], starts[6];
      std::vector<float> fl(d.size() / 4 + 8); std::vector<long long> in(d.size() + 8), sp(2 * (d.size() / 2 + 8));
      if (c2d_example_parse(d.data(), (long long)d.size(), keys, 6, kinds, counts, starts, fl.data(), (long long)fl.size(), in.data(), (long long)in.size(), sp.data(), (long long)sp.size() / 2) == 0) ++oke; }
    { std::vector<uint8_t> d = mutate(rec); long long pos = 0, off, len;
      for (int k = 0; k < 100; ++k) { long long n = c2d_tfrecord_next(d.data(), (long long)d.size(), pos, &off, &len, (int)(it & 1)); if (n <= 0) break; volatile uint8_t s = 0; for (long long q = 0; q < len; ++q) s ^= d[off + q]; pos = n; ++okr; } }
  }
  printf("iterations %ld: jpeg ok %ld, example ok %ld, records read %ld — no sanitizer report\n", iters, okj, oke, okr);
  return 0;
}
