// Native (host C++) input path of the Cap2Det reader — SURVEY.md §8f row f1.
//
// The reference reads TFRecord files of tf.Example protos through tf.data's C++ runtime
// (readers/cap2det_reader.py:31-139: TFRecordDataset -> tf.parse_single_example ->
// tf.image.decode_jpeg); the files are written by dataset-tools/create_pascal_tf_record.py:
// 147-196 / create_coco_tf_record.py:197-242.  This file restates, from their published formats,
// what those third-party pieces do on the host:
//   * TFRecord framing: u64 length | masked crc32c(length) | payload | masked crc32c(payload),
//     mask(c) = rotr(c, 15) + 0xa282ead8, CRC-32C (Castagnoli) — tensorflow/core/lib/io/
//     record_reader.cc, lib/hash/crc32c.h;
//   * tf.Example wire format (protobuf): Example{1: Features{1: map<string, Feature>}} with
//     Feature{1: BytesList | 2: FloatList | 3: Int64List}, packed or unpacked repeated scalars;
//   * baseline / extended-sequential AND progressive (SOF2, multi-scan) Huffman JPEG decoding with libjpeg's default decompression
//     choices, which tf.image.decode_jpeg uses (dct_method "" = JDCT_ISLOW, fancy_upscaling =
//     True): the integer "slow-but-accurate" IDCT of jidctint.c, triangle-filter ("fancy")
//     chroma upsampling of jdsample.c and the 16-bit fixed-point YCbCr->RGB of jdcolor.c, so the
//     decoded pixels are bit-identical to libjpeg(-turbo)'s (tests pin this against the system
//     libjpeg-turbo through Pillow, progressive files included; arithmetic-coded and lossless
//     files return C2D_ERR_UNSUPPORTED);
//   * tf.strings.to_hash_bucket (the reader's shard filter, readers/cap2det_reader.py:201-211):
//     TensorFlow's Hash64 (a MurmurHash64A variant, seed 0xDECAFCAFFE) modulo the bucket count.
// Entropy decoding is inherently serial, so it stays on host threads (ctypes releases the GIL
// around these calls); everything at pixel rate after it (flip, resize, padding, batch rescale)
// runs on the GPU (preprocess.hip).
#include <stdint.h>
#include <string.h>

#include "../../include/cap2det_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------
// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8 tables built on first use
// ---------------------------------------------------------------------------------------------
uint32_t g_crc[8][256];
bool g_crc_ready = false;

void crc_init() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
    g_crc[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc[t][i] = (g_crc[t - 1][i] >> 8) ^ g_crc[0][g_crc[t - 1][i] & 0xff];
  g_crc_ready = true;
}

uint32_t crc32c(const uint8_t* p, size_t n) {
  if (!g_crc_ready) crc_init();
  uint32_t c = 0xffffffffu;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = g_crc[7][lo & 0xff] ^ g_crc[6][(lo >> 8) & 0xff] ^ g_crc[5][(lo >> 16) & 0xff] ^
        g_crc[4][lo >> 24] ^ g_crc[3][hi & 0xff] ^ g_crc[2][(hi >> 8) & 0xff] ^
        g_crc[1][(hi >> 16) & 0xff] ^ g_crc[0][hi >> 24];
    p += 8; n -= 8;
  }
  while (n--) c = (c >> 8) ^ g_crc[0][(c ^ *p++) & 0xff];
  return c ^ 0xffffffffu;
}

inline uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

// ---------------------------------------------------------------------------------------------
// protobuf wire helpers
// ---------------------------------------------------------------------------------------------
struct Cursor {
  const uint8_t* p;
  const uint8_t* end;
  bool ok;
};

uint64_t varint(Cursor& c) {
  uint64_t v = 0;
  for (int shift = 0; shift < 64; shift += 7) {
    if (c.p >= c.end) { c.ok = false; return 0; }
    const uint8_t b = *c.p++;
    v |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) return v;
  }
  c.ok = false;
  return 0;
}

// reads a length-delimited field body as a sub-cursor
Cursor sub(Cursor& c) {
  const uint64_t n = varint(c);
  Cursor s = {c.p, c.p, c.ok};
  if (!c.ok || n > (uint64_t)(c.end - c.p)) { c.ok = false; s.ok = false; return s; }
  s.end = c.p + n;
  c.p += n;
  return s;
}

bool skip(Cursor& c, int wire) {
  switch (wire) {
    case 0: varint(c); return c.ok;
    case 1: if (c.end - c.p < 8) return c.ok = false; c.p += 8; return true;
    case 2: sub(c); return c.ok;
    case 5: if (c.end - c.p < 4) return c.ok = false; c.p += 4; return true;
    default: return c.ok = false;
  }
}

}  // namespace

extern "C" unsigned int c2d_crc32c(const void* data, long long n) {
  return crc32c((const uint8_t*)data, (size_t)(n > 0 ? n : 0));
}

extern "C" unsigned int c2d_masked_crc32c(const void* data, long long n) {
  return mask_crc(crc32c((const uint8_t*)data, (size_t)(n > 0 ? n : 0)));
}

// Next record of a TFRecord byte buffer starting at `pos`.  Returns the position after the
// record (> pos), 0 at a clean end of buffer, or a negative C2D_ERR_* (truncated / bad CRC).
extern "C" long long c2d_tfrecord_next(const uint8_t* buf, long long size, long long pos,
                                       long long* payload_off, long long* payload_len,
                                       int verify_crc) {
  if (!buf || !payload_off || !payload_len || pos < 0 || pos > size) return C2D_ERR_INVALID_ARG;
  if (pos == size) return 0;
  if (size - pos < 16) return C2D_ERR_DATA;      // length + its CRC + payload CRC
  uint64_t len;
  uint32_t lcrc;
  memcpy(&len, buf + pos, 8);
  memcpy(&lcrc, buf + pos + 8, 4);
  if (verify_crc && mask_crc(crc32c(buf + pos, 8)) != lcrc) return C2D_ERR_DATA;
  if (len > (uint64_t)(size - pos - 16)) return C2D_ERR_DATA;
  if (verify_crc) {
    uint32_t dcrc;
    memcpy(&dcrc, buf + pos + 12 + len, 4);
    if (mask_crc(crc32c(buf + pos + 12, (size_t)len)) != dcrc) return C2D_ERR_DATA;
  }
  *payload_off = pos + 12;
  *payload_len = (long long)len;
  return pos + 16 + (long long)len;
}

// Frames `payload` as one TFRecord into out (>= n + 16 bytes); returns the bytes written.
extern "C" long long c2d_tfrecord_frame(const uint8_t* payload, long long n, uint8_t* out) {
  if (!out || n < 0 || (n > 0 && !payload)) return C2D_ERR_INVALID_ARG;
  const uint64_t len = (uint64_t)n;
  memcpy(out, &len, 8);
  const uint32_t lcrc = mask_crc(crc32c(out, 8));
  memcpy(out + 8, &lcrc, 4);
  if (n) memcpy(out + 12, payload, (size_t)n);
  const uint32_t dcrc = mask_crc(crc32c(out + 12, (size_t)n));
  memcpy(out + 12 + n, &dcrc, 4);
  return n + 16;
}

// Parses one serialized tf.Example.  For each of the `nkeys` requested feature names fills
// kinds[k] (0 absent / empty kind, 1 bytes, 2 float, 3 int64), counts[k] and starts[k] (index
// of the feature's first value in the arena of its kind).  Bytes values are returned as
// (offset, length) pairs into `rec` (spans[2*i], spans[2*i+1]).  A feature that appears twice
// keeps its last occurrence (protobuf map semantics).  Returns C2D_OK, C2D_ERR_DATA on a
// malformed record, C2D_ERR_WORKSPACE when an arena is too small.
extern "C" int c2d_example_parse(const uint8_t* rec, long long len, const char* const* keys,
                                 int nkeys, int* kinds, long long* counts, long long* starts,
                                 float* floats, long long float_cap, long long* ints,
                                 long long int_cap, long long* spans, long long span_cap) {
  if (!rec || len < 0 || !keys || nkeys < 0 || !kinds || !counts || !starts)
    return C2D_ERR_INVALID_ARG;
  for (int k = 0; k < nkeys; ++k) { kinds[k] = 0; counts[k] = 0; starts[k] = 0; }
  long long nf = 0, ni = 0, ns = 0;
  Cursor ex = {rec, rec + len, true};
  while (ex.ok && ex.p < ex.end) {
    const uint64_t tag = varint(ex);
    if (!ex.ok) break;
    if ((tag >> 3) != 1 || (tag & 7) != 2) { if (!skip(ex, (int)(tag & 7))) break; continue; }
    Cursor feats = sub(ex);                       // Features
    while (feats.ok && feats.p < feats.end) {
      const uint64_t t2 = varint(feats);
      if (!feats.ok) break;
      if ((t2 >> 3) != 1 || (t2 & 7) != 2) { if (!skip(feats, (int)(t2 & 7))) break; continue; }
      Cursor entry = sub(feats);                  // map entry {1: key, 2: Feature}
      const uint8_t* kptr = nullptr; size_t klen = 0;
      Cursor fval = {nullptr, nullptr, false};
      while (entry.ok && entry.p < entry.end) {
        const uint64_t t3 = varint(entry);
        if (!entry.ok) break;
        if ((t3 & 7) == 2 && (t3 >> 3) == 1) { Cursor s = sub(entry); kptr = s.p; klen = (size_t)(s.end - s.p); }
        else if ((t3 & 7) == 2 && (t3 >> 3) == 2) { fval = sub(entry); }
        else if (!skip(entry, (int)(t3 & 7))) break;
      }
      if (!entry.ok) { feats.ok = false; break; }
      int which = -1;
      for (int k = 0; k < nkeys && kptr; ++k)
        if (strlen(keys[k]) == klen && memcmp(keys[k], kptr, klen) == 0) { which = k; break; }
      if (which < 0 || !fval.ok) continue;
      kinds[which] = 0; counts[which] = 0;
      while (fval.ok && fval.p < fval.end) {     // Feature oneof
        const uint64_t t4 = varint(fval);
        if (!fval.ok) break;
        const int field = (int)(t4 >> 3);
        if ((t4 & 7) != 2 || field < 1 || field > 3) { if (!skip(fval, (int)(t4 & 7))) break; continue; }
        Cursor list = sub(fval);
        kinds[which] = field; counts[which] = 0;
        starts[which] = field == 1 ? ns : (field == 2 ? nf : ni);
        while (list.ok && list.p < list.end) {
          const uint64_t t5 = varint(list);
          if (!list.ok) break;
          if ((t5 >> 3) != 1) { if (!skip(list, (int)(t5 & 7))) break; continue; }
          const int wire = (int)(t5 & 7);
          if (field == 1 && wire == 2) {                               // bytes value
            Cursor s = sub(list);
            if (!list.ok) break;
            if (ns >= span_cap || !spans) return C2D_ERR_WORKSPACE;
            spans[2 * ns] = (long long)(s.p - rec); spans[2 * ns + 1] = (long long)(s.end - s.p);
            ++ns; ++counts[which];
          } else if (field == 2 && wire == 2) {                        // packed floats
            Cursor s = sub(list);
            if (!list.ok || ((s.end - s.p) & 3)) { list.ok = false; break; }
            const long long m = (s.end - s.p) / 4;
            if (nf + m > float_cap || !floats) return C2D_ERR_WORKSPACE;
            memcpy(floats + nf, s.p, (size_t)m * 4);
            nf += m; counts[which] += m;
          } else if (field == 2 && wire == 5) {                        // unpacked float
            if (list.end - list.p < 4) { list.ok = false; break; }
            if (nf >= float_cap || !floats) return C2D_ERR_WORKSPACE;
            memcpy(floats + nf, list.p, 4); list.p += 4;
            ++nf; ++counts[which];
          } else if (field == 3 && wire == 2) {                        // packed int64 varints
            Cursor s = sub(list);
            if (!list.ok) break;
            while (s.ok && s.p < s.end) {
              const uint64_t v = varint(s);
              if (!s.ok) break;
              if (ni >= int_cap || !ints) return C2D_ERR_WORKSPACE;
              ints[ni++] = (long long)v; ++counts[which];
            }
            if (!s.ok) { list.ok = false; break; }
          } else if (field == 3 && wire == 0) {                        // unpacked int64
            const uint64_t v = varint(list);
            if (!list.ok) break;
            if (ni >= int_cap || !ints) return C2D_ERR_WORKSPACE;
            ints[ni++] = (long long)v; ++counts[which];
          } else if (!skip(list, wire)) {
            break;
          }
        }
        if (!list.ok) { fval.ok = false; break; }
      }
      if (!fval.ok) { feats.ok = false; break; }
    }
    if (!feats.ok) { ex.ok = false; break; }
  }
  return ex.ok ? C2D_OK : C2D_ERR_DATA;
}

// tf.strings.to_hash_bucket: TensorFlow's Hash64(data, n, seed = 0xDECAFCAFFE) % num_buckets.
extern "C" unsigned long long c2d_tf_hash64(const void* data, long long n) {
  const uint64_t m = 0xc6a4a7935bd1e995ull;
  const int r = 47;
  const uint8_t* p = (const uint8_t*)data;
  size_t len = (size_t)(n > 0 ? n : 0);
  uint64_t h = 0xDECAFCAFFEull ^ (len * m);
  while (len >= 8) {
    uint64_t k;
    memcpy(&k, p, 8);
    p += 8; len -= 8;
    k *= m; k ^= k >> r; k *= m;
    h ^= k; h *= m;
  }
  switch (len) {
    case 7: h ^= (uint64_t)p[6] << 48;  // fallthrough
    case 6: h ^= (uint64_t)p[5] << 40;  // fallthrough
    case 5: h ^= (uint64_t)p[4] << 32;  // fallthrough
    case 4: h ^= (uint64_t)p[3] << 24;  // fallthrough
    case 3: h ^= (uint64_t)p[2] << 16;  // fallthrough
    case 2: h ^= (uint64_t)p[1] << 8;   // fallthrough
    case 1: h ^= (uint64_t)p[0]; h *= m;
  }
  h ^= h >> r; h *= m; h ^= h >> r;
  return h;
}

// =============================================================================================
// JPEG (baseline / extended sequential, Huffman, 8-bit) -> interleaved RGB
// =============================================================================================
namespace {

struct Huff {
  // canonical code tables: for code length l (1..16): mincode[l], maxcode[l] (-1 = none),
  // valptr[l]; plus a 9-bit lookahead table (length << 8 | symbol, 0 = longer code)
  int mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  uint16_t look[512];
  bool present;
};

struct Comp {
  int id, h, v, tq, td, ta;
  int dw, dh;            // real downsampled size
  int pw, ph;            // padded plane size (whole MCUs)
  int dc_pred;
  uint8_t* plane;
};

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint32_t buf;
  int bits;
  bool marker;           // a marker (other than stuffed 0xFF00) was reached: feed zeros
  bool eof;              // the data ended inside entropy-coded bits (truncated / corrupted file)
  int pad;               // zero bytes fed after a marker was reached (<= 4 in a valid stream)
};

inline void fill(BitReader& br) {
  while (br.bits <= 24) {
    uint32_t byte = 0;
    if (!br.marker && br.p < br.end) {
      byte = *br.p;
      if (byte == 0xff) {
        if (br.p + 1 < br.end && br.p[1] == 0x00) { br.p += 2; }
        else { br.marker = true; byte = 0; }
      } else {
        ++br.p;
      }
    } else if (!br.marker) {
      br.eof = true;
    } else if (++br.pad > 16) {
      br.eof = true;     // the scan wants far more bits than it holds: stop instead of decoding
    }                    // zeros for the rest of a (possibly huge, corrupted-size) image
    br.buf |= byte << (24 - br.bits);
    br.bits += 8;
  }
}

inline int get_bits(BitReader& br, int n) {
  if (n == 0) return 0;
  if (br.bits < n) fill(br);
  const int v = (int)(br.buf >> (32 - n));
  br.buf <<= n; br.bits -= n;
  return v;
}

inline int decode_sym(BitReader& br, const Huff& h) {
  if (br.bits < 16) fill(br);
  const uint16_t lk = h.look[br.buf >> 23];
  if (lk) {
    const int l = lk >> 8;
    br.buf <<= l; br.bits -= l;
    return lk & 0xff;
  }
  int code = (int)(br.buf >> 23), l = 9;
  uint32_t rest = br.buf << 9;
  while (l <= 16) {
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) break;
    code = (code << 1) | (int)(rest >> 31);
    rest <<= 1;
    ++l;
  }
  if (l > 16) return -1;
  br.buf <<= l; br.bits -= l;
  return h.vals[h.valptr[l] + code - h.mincode[l]];
}

inline int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                             12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                             58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
inline int clampc(long long v) { return (int)(v > (1 << 20) ? (1 << 20) : (v < -(1 << 20) ? -(1 << 20) : v)); }

// jidctint.c (jpeg_idct_islow): LL&M integer IDCT, CONST_BITS = 13, PASS1_BITS = 2.
// Intermediates are 64-bit: identical results on valid streams (libjpeg's 32-bit arithmetic
// never overflows there) and no signed overflow on corrupted ones (coefficients are clamped to
// +-2^20 by the caller).
typedef long long idct_t;
inline uint8_t clamp8l(idct_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
void idct_islow(const int* coef /* dequantized, natural order */, uint8_t* out, int stride) {
  const int CB = 13, P1 = 2;
  const idct_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633,
            F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
  idct_t ws[64];
  for (int c = 0; c < 8; ++c) {
    const int* in = coef + c;
    idct_t* w = ws + c;
    if (in[8] == 0 && in[16] == 0 && in[24] == 0 && in[32] == 0 && in[40] == 0 && in[48] == 0 &&
        in[56] == 0) {
      const idct_t dc = (idct_t)in[0] * (1 << P1);
      for (int r = 0; r < 8; ++r) w[8 * r] = dc;
      continue;
    }
    idct_t z2 = in[16], z3 = in[48];
    idct_t z1 = (z2 + z3) * F0541;
    idct_t tmp2 = z1 + z3 * (-F1847);
    idct_t tmp3 = z1 + z2 * F0765;
    z2 = in[0]; z3 = in[32];
    idct_t tmp0 = (z2 + z3) * (1 << CB);
    idct_t tmp1 = (z2 - z3) * (1 << CB);
    const idct_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[56]; tmp1 = in[40]; tmp2 = in[24]; tmp3 = in[8];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    idct_t z4 = tmp1 + tmp3;
    const idct_t z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const int sh = CB - P1;
    const idct_t rnd = 1 << (sh - 1);
    w[0] = (tmp10 + tmp3 + rnd) >> sh;  w[56] = (tmp10 - tmp3 + rnd) >> sh;
    w[8] = (tmp11 + tmp2 + rnd) >> sh;  w[48] = (tmp11 - tmp2 + rnd) >> sh;
    w[16] = (tmp12 + tmp1 + rnd) >> sh; w[40] = (tmp12 - tmp1 + rnd) >> sh;
    w[24] = (tmp13 + tmp0 + rnd) >> sh; w[32] = (tmp13 - tmp0 + rnd) >> sh;
  }
  const int sh = CB + P1 + 3;
  const idct_t rnd = 1 << (sh - 1);
  for (int r = 0; r < 8; ++r) {
    const idct_t* w = ws + 8 * r;
    uint8_t* o = out + r * stride;
    if (w[1] == 0 && w[2] == 0 && w[3] == 0 && w[4] == 0 && w[5] == 0 && w[6] == 0 && w[7] == 0) {
      const uint8_t dc = clamp8l(((w[0] + (1 << (P1 + 2))) >> (P1 + 3)) + 128);
      for (int c = 0; c < 8; ++c) o[c] = dc;
      continue;
    }
    idct_t z2 = w[2], z3 = w[6];
    idct_t z1 = (z2 + z3) * F0541;
    idct_t tmp2 = z1 + z3 * (-F1847);
    idct_t tmp3 = z1 + z2 * F0765;
    idct_t tmp0 = (w[0] + w[4]) * (1 << CB);
    idct_t tmp1 = (w[0] - w[4]) * (1 << CB);
    const idct_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    idct_t z4 = tmp1 + tmp3;
    const idct_t z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    o[0] = clamp8l(((tmp10 + tmp3 + rnd) >> sh) + 128); o[7] = clamp8l(((tmp10 - tmp3 + rnd) >> sh) + 128);
    o[1] = clamp8l(((tmp11 + tmp2 + rnd) >> sh) + 128); o[6] = clamp8l(((tmp11 - tmp2 + rnd) >> sh) + 128);
    o[2] = clamp8l(((tmp12 + tmp1 + rnd) >> sh) + 128); o[5] = clamp8l(((tmp12 - tmp1 + rnd) >> sh) + 128);
    o[3] = clamp8l(((tmp13 + tmp0 + rnd) >> sh) + 128); o[4] = clamp8l(((tmp13 - tmp0 + rnd) >> sh) + 128);
  }
}

struct Header {
  int width, height, ncomp, maxh, maxv, restart;
  bool progressive, have_sof;
  Comp comp[3];
  uint16_t quant[4][64];
  bool have_q[4];
  Huff dc[4], ac[4];
  const uint8_t* scan;     // entropy-coded data of the first scan
  int scan_ncomp;
  long long sos_pos;       // offset of the first SOS marker's length field
  bool single_scan;        // sequential, one interleaved scan of all components (fast path)
};

bool build_huff(Huff& h, const uint8_t* counts, const uint8_t* vals, int nvals) {
  int code = 0, k = 0;
  memcpy(h.vals, vals, (size_t)nvals);
  memset(h.look, 0, sizeof(h.look));
  for (int l = 1; l <= 16; ++l) {
    h.valptr[l] = k;
    h.mincode[l] = code;
    if (counts[l - 1]) {
      for (int i = 0; i < counts[l - 1]; ++i, ++k, ++code) {
        if (code >= (1 << l)) return false;       // over-subscribed code lengths
        if (l <= 9)
          for (int f = 0; f < (1 << (9 - l)); ++f)
            h.look[(code << (9 - l)) | f] = (uint16_t)((l << 8) | vals[k]);
      }
      h.maxcode[l] = code - 1;
    } else {
      h.maxcode[l] = -1;
    }
    if (code > (1 << l)) return false;
    code <<= 1;
  }
  h.maxcode[17] = 0x7fffffff;
  h.present = true;
  return k == nvals;
}

// DQT / DHT / DRI segments (they may also appear between the scans of a multi-scan file).
int parse_tables(int m, const uint8_t* s, int sl, Header& hd) {
  if (m == 0xdb) {
    int q = 0;
    while (q < sl) {
      const int pq = s[q] >> 4, tq = s[q] & 15;
      if (tq > 3 || q + 1 + (pq ? 128 : 64) > sl) return C2D_ERR_DATA;
      for (int i = 0; i < 64; ++i)
        hd.quant[tq][kZigzag[i]] = pq ? (uint16_t)((s[q + 1 + 2 * i] << 8) | s[q + 2 + 2 * i]) : s[q + 1 + i];
      hd.have_q[tq] = true;
      q += 1 + (pq ? 128 : 64);
    }
  } else if (m == 0xc4) {
    int q = 0;
    while (q < sl) {
      if (q + 17 > sl) return C2D_ERR_DATA;
      const int tc = s[q] >> 4, th = s[q] & 15;
      int nv = 0;
      for (int i = 0; i < 16; ++i) nv += s[q + 1 + i];
      if (th > 3 || tc > 1 || nv > 256 || q + 17 + nv > sl) return C2D_ERR_DATA;
      if (!build_huff(tc ? hd.ac[th] : hd.dc[th], s + q + 1, s + q + 17, nv)) return C2D_ERR_DATA;
      q += 17 + nv;
    }
  } else if (m == 0xdd) {
    if (sl < 2) return C2D_ERR_DATA;
    hd.restart = (s[0] << 8) | s[1];
  }
  return C2D_OK;
}

int parse_header(const uint8_t* d, long long n, Header& hd) {
  memset(&hd, 0, sizeof(hd));
  if (n < 4 || d[0] != 0xff || d[1] != 0xd8) return C2D_ERR_DATA;
  long long p = 2;
  while (p + 4 <= n) {
    if (d[p] != 0xff) return C2D_ERR_DATA;
    while (p < n && d[p] == 0xff) ++p;          // fill bytes
    if (p >= n) return C2D_ERR_DATA;
    const int m = d[p++];
    if (m == 0xd8 || m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
    if (m == 0xd9) return C2D_ERR_DATA;         // EOI before a scan
    if (p + 2 > n) return C2D_ERR_DATA;
    const int len = (d[p] << 8) | d[p + 1];
    if (len < 2 || p + len > n) return C2D_ERR_DATA;
    const uint8_t* s = d + p + 2;
    const int sl = len - 2;
    if (m == 0xc0 || m == 0xc1 || m == 0xc2) {
      if (sl < 6 || s[0] != 8) return m == 0xc2 ? C2D_ERR_UNSUPPORTED : C2D_ERR_UNSUPPORTED;
      hd.progressive = m == 0xc2;
      hd.height = (s[1] << 8) | s[2]; hd.width = (s[3] << 8) | s[4]; hd.ncomp = s[5];
      if ((hd.ncomp != 1 && hd.ncomp != 3) || sl < 6 + 3 * hd.ncomp || hd.width <= 0 || hd.height <= 0)
        return C2D_ERR_UNSUPPORTED;
      for (int i = 0; i < hd.ncomp; ++i) {
        Comp& c = hd.comp[i];
        c.id = s[6 + 3 * i]; c.h = s[7 + 3 * i] >> 4; c.v = s[7 + 3 * i] & 15; c.tq = s[8 + 3 * i];
        if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return C2D_ERR_UNSUPPORTED;
        // a single-component scan is never interleaved: its MCU is one 8x8 block whatever the
        // declared sampling factors (ITU T.81 A.2.2)
        if (hd.ncomp == 1) { c.h = 1; c.v = 1; }
        if (c.h > hd.maxh) hd.maxh = c.h;
        if (c.v > hd.maxv) hd.maxv = c.v;
      }
      hd.have_sof = true;
    } else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
      return C2D_ERR_UNSUPPORTED;               // lossless / hierarchical / arithmetic
    } else if (m == 0xdb || m == 0xc4 || m == 0xdd) {
      const int rc = parse_tables(m, s, sl, hd);
      if (rc) return rc;
    } else if (m == 0xda) {
      if (!hd.have_sof || sl < 1) return C2D_ERR_DATA;
      const int ns = s[0];
      if (ns < 1 || ns > hd.ncomp || sl < 1 + 2 * ns + 3) return C2D_ERR_DATA;
      hd.sos_pos = p;
      hd.single_scan = !hd.progressive && ns == hd.ncomp;
      for (int i = 0; i < ns && hd.single_scan; ++i) {
        int ci = -1;
        for (int j = 0; j < hd.ncomp; ++j) if (hd.comp[j].id == s[1 + 2 * i]) ci = j;
        if (ci != i) { hd.single_scan = false; break; }
        hd.comp[ci].td = s[2 + 2 * i] >> 4; hd.comp[ci].ta = s[2 + 2 * i] & 15;
        if (hd.comp[ci].td > 3 || hd.comp[ci].ta > 3) return C2D_ERR_DATA;
      }
      hd.scan = d + p + len;
      hd.scan_ncomp = ns;
      {
        // every 8x8 block costs at least one bit (its DC Huffman code): a header announcing more
        // blocks than the file has bits is corrupt — refuse before anyone sizes buffers by it
        const long long mcux = (hd.width + 8 * hd.maxh - 1) / (8 * hd.maxh);
        const long long mcuy = (hd.height + 8 * hd.maxv - 1) / (8 * hd.maxv);
        long long per_mcu = 0;
        for (int i = 0; i < hd.ncomp; ++i) per_mcu += hd.comp[i].h * hd.comp[i].v;
        if (mcux * mcuy * per_mcu > 8 * n) return C2D_ERR_DATA;
      }
      return C2D_OK;
    }
    p += len;
  }
  return C2D_ERR_DATA;
}

// ---------------------------------------------------------------------------------------------
// Multi-scan files: progressive (SOF2; jdphuff.c: DC / AC, first / refinement scans with EOB runs)
// and sequential files whose components come in separate scans.  Every scan is entropy-decoded
// into a coefficient image [component][block row][block column][64] (natural order, NOT
// dequantised); the caller then dequantises and runs the same IDCT / upsampling / colour
// conversion as the single-scan path, so a complete file decodes to the same pixels as
// libjpeg-turbo (its block smoothing only acts on files whose scans are missing).
// ---------------------------------------------------------------------------------------------
struct ScanComp { int ci, td, ta; };

inline void restart_sync(BitReader& br, int& next_rst, bool& ok) {
  br.bits = 0; br.buf = 0;
  const uint8_t* q = br.p;
  while (q + 1 < br.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) ++q;
  if (q + 1 >= br.end || q[1] != 0xd0 + next_rst) { ok = false; return; }
  br.p = q + 2; br.marker = false; br.pad = 0;
  next_rst = (next_rst + 1) & 7;
}

// one block of one scan; returns false on a corrupted stream
inline bool decode_block_scan(BitReader& br, const Header& hd, bool progressive, const Huff* dct,
                              const Huff* act, int ss, int se, int ah, int al, int& dc_pred,
                              int& eobrun, short* blk) {
  if (!progressive) {                       // sequential: the whole block
    int t = decode_sym(br, *dct);
    if (t < 0 || t > 11) return false;
    dc_pred += t ? extend(get_bits(br, t), t) : 0;
    if (dc_pred > 32767) dc_pred = 32767;
    if (dc_pred < -32768) dc_pred = -32768;
    blk[0] = (short)dc_pred;
    for (int k = 1; k < 64;) {
      const int rs = decode_sym(br, *act);
      if (rs < 0) return false;
      const int r = rs >> 4, s = rs & 15;
      if (s == 0) {
        if (r == 15) { k += 16; continue; }
        break;
      }
      k += r;
      if (k > 63) return false;
      blk[kZigzag[k]] = (short)extend(get_bits(br, s), s);
      ++k;
    }
    return true;
  }
  if (ss == 0) {
    if (ah == 0) {                          // DC first
      int t = decode_sym(br, *dct);
      if (t < 0 || t > 11) return false;
      dc_pred += t ? extend(get_bits(br, t), t) : 0;
      if (dc_pred > 32767) dc_pred = 32767;
      if (dc_pred < -32768) dc_pred = -32768;
      blk[0] = (short)((unsigned)dc_pred << al);
    } else if (get_bits(br, 1)) {           // DC refinement
      blk[0] = (short)(blk[0] | (1 << al));
    }
    return true;
  }
  if (ah == 0) {                            // AC first
    if (eobrun > 0) { --eobrun; return true; }
    for (int k = ss; k <= se; ++k) {
      const int rs = decode_sym(br, *act);
      if (rs < 0) return false;
      const int r = rs >> 4, s = rs & 15;
      if (s) {
        k += r;
        if (k > 63) return false;
        blk[kZigzag[k]] = (short)((unsigned)extend(get_bits(br, s), s) << al);
      } else if (r == 15) {
        k += 15;
      } else {
        eobrun = 1 << r;
        if (r) eobrun += get_bits(br, r);
        --eobrun;
        break;
      }
    }
    return true;
  }
  // AC refinement
  const int p1 = 1 << al, m1 = -(1 << al);
  int k = ss;
  if (eobrun == 0) {
    for (; k <= se; ++k) {
      const int rs = decode_sym(br, *act);
      if (rs < 0) return false;
      int r = rs >> 4, s = rs & 15;
      if (s) {
        if (s != 1) return false;
        s = get_bits(br, 1) ? p1 : m1;
      } else if (r != 15) {
        eobrun = 1 << r;
        if (r) eobrun += get_bits(br, r);
        break;
      }
      // pass over already-nonzero coefficients (each takes a correction bit) and r zeros
      do {
        short* c = blk + kZigzag[k];
        if (*c != 0) {
          if (get_bits(br, 1) && (*c & p1) == 0) *c = (short)(*c + (*c >= 0 ? p1 : m1));
        } else if (--r < 0) {
          break;
        }
        ++k;
      } while (k <= se);
      if (s) {
        if (k > 63) return false;
        blk[kZigzag[k]] = (short)s;
      }
    }
  }
  if (eobrun > 0) {
    for (; k <= se; ++k) {
      short* c = blk + kZigzag[k];
      if (*c != 0 && get_bits(br, 1) && (*c & p1) == 0) *c = (short)(*c + (*c >= 0 ? p1 : m1));
    }
    --eobrun;
  }
  return true;
}

// Decodes every scan from the first SOS to EOI into coef[ci] ([ph/8][pw/8][64] shorts, zeroed by
// the caller).  Components' dw/dh/pw/ph are set.
int decode_scans(const uint8_t* d, long long n, Header& hd, short* const* coef) {
  long long p = hd.sos_pos;                 // at the length field of an SOS segment
  const int mcuw = 8 * hd.maxh, mcuh = 8 * hd.maxv;
  const int mcux = (hd.width + mcuw - 1) / mcuw, mcuy = (hd.height + mcuh - 1) / mcuh;
  int scans = 0;
  for (;;) {
    if (p + 2 > n) return C2D_ERR_DATA;
    const int len = (d[p] << 8) | d[p + 1];
    if (len < 2 || p + len > n) return C2D_ERR_DATA;
    const uint8_t* s = d + p + 2;
    const int sl = len - 2;
    const int ns = sl >= 1 ? s[0] : 0;
    if (ns < 1 || ns > hd.ncomp || sl < 1 + 2 * ns + 3) return C2D_ERR_DATA;
    ScanComp sc[3];
    for (int i = 0; i < ns; ++i) {
      sc[i].ci = -1;
      for (int j = 0; j < hd.ncomp; ++j) if (hd.comp[j].id == s[1 + 2 * i]) sc[i].ci = j;
      sc[i].td = s[2 + 2 * i] >> 4; sc[i].ta = s[2 + 2 * i] & 15;
      if (sc[i].ci < 0 || sc[i].td > 3 || sc[i].ta > 3) return C2D_ERR_DATA;
      for (int j = 0; j < i; ++j) if (sc[j].ci == sc[i].ci) return C2D_ERR_DATA;
    }
    int ss = s[1 + 2 * ns], se = s[2 + 2 * ns];
    int ah = s[3 + 2 * ns] >> 4, al = s[3 + 2 * ns] & 15;
    if (hd.progressive) {
      if (ss > se || se > 63 || ah > 13 || al > 13 || (ss == 0 && se != 0) || (ss > 0 && ns != 1))
        return C2D_ERR_DATA;
    } else {
      ss = 0; se = 63; ah = 0; al = 0;
    }
    for (int i = 0; i < ns; ++i) {
      const bool need_dc = !hd.progressive || (ss == 0 && ah == 0);
      const bool need_ac = !hd.progressive || ss > 0;
      if ((need_dc && !hd.dc[sc[i].td].present) || (need_ac && !hd.ac[sc[i].ta].present))
        return C2D_ERR_DATA;
    }
    if (++scans > 1000) return C2D_ERR_DATA;
    BitReader br = {d + p + len, d + n, 0, 0, false};
    int pred[3] = {0, 0, 0};
    int eobrun = 0, rst_left = hd.restart, next_rst = 0;
    bool ok = true;
    if (ns == 1) {
      // non-interleaved: the component's own blocks in raster order (no MCU padding)
      Comp& c = hd.comp[sc[0].ci];
      const int bw = (c.dw + 7) / 8, bh = (c.dh + 7) / 8, pbw = c.pw / 8;
      for (int by = 0; by < bh && ok && !br.eof; ++by)
        for (int bx = 0; bx < bw; ++bx) {
          if (hd.restart && rst_left == 0) {
            restart_sync(br, next_rst, ok);
            if (!ok) break;
            rst_left = hd.restart; pred[0] = 0; eobrun = 0;
          }
          ok = decode_block_scan(br, hd, hd.progressive, &hd.dc[sc[0].td], &hd.ac[sc[0].ta], ss, se,
                                 ah, al, pred[0], eobrun,
                                 coef[sc[0].ci] + ((size_t)by * pbw + bx) * 64);
          if (!ok) break;
          if (hd.restart) --rst_left;
        }
    } else {
      for (int my = 0; my < mcuy && ok && !br.eof; ++my)
        for (int mx = 0; mx < mcux && ok; ++mx) {
          if (hd.restart && rst_left == 0) {
            restart_sync(br, next_rst, ok);
            if (!ok) break;
            rst_left = hd.restart; pred[0] = pred[1] = pred[2] = 0; eobrun = 0;
          }
          for (int i = 0; i < ns && ok; ++i) {
            Comp& c = hd.comp[sc[i].ci];
            const int pbw = c.pw / 8;
            for (int by = 0; by < c.v && ok; ++by)
              for (int bx = 0; bx < c.h && ok; ++bx)
                ok = decode_block_scan(br, hd, hd.progressive, &hd.dc[sc[i].td], &hd.ac[sc[i].ta],
                                       ss, se, ah, al, pred[i], eobrun,
                                       coef[sc[i].ci] + ((size_t)(my * c.v + by) * pbw + mx * c.h + bx) * 64);
          }
          if (hd.restart) --rst_left;
        }
    }
    if (!ok || br.eof) return C2D_ERR_DATA;   // (truncated: tf.image.decode_jpeg fails too)
    // ---- markers up to the next scan ----
    long long q = br.p - d;
    for (;;) {
      while (q + 1 < n && !(d[q] == 0xff && d[q + 1] != 0x00 && d[q + 1] != 0xff &&
                            !(d[q + 1] >= 0xd0 && d[q + 1] <= 0xd7))) ++q;
      if (q + 1 >= n) return C2D_ERR_DATA;  // no EOI: the remaining scans are missing
      const int m = d[q + 1];
      q += 2;
      if (m == 0xd9) return C2D_OK;
      if (q + 2 > n) return C2D_ERR_DATA;
      const int l2 = (d[q] << 8) | d[q + 1];
      if (l2 < 2 || q + l2 > n) return C2D_ERR_DATA;
      if (m == 0xda) { p = q; break; }
      const int rc = parse_tables(m, d + q + 2, l2 - 2, hd);
      if (rc) return rc;
      q += l2;
    }
  }
}

}  // namespace

extern "C" int c2d_jpeg_info(const uint8_t* data, long long n, int* height, int* width,
                             int* components) {
  if (!data || !height || !width || !components) return C2D_ERR_INVALID_ARG;
  Header hd;
  const int rc = parse_header(data, n, hd);
  if (hd.have_sof) { *height = hd.height; *width = hd.width; *components = hd.ncomp; }
  return rc;
}

// Decodes to interleaved RGB u8 [height][width][3] (grayscale is replicated, as
// tf.image.decode_jpeg(channels=3) does).  workspace >= c2d_jpeg_workspace_bytes(...).
extern "C" long long c2d_jpeg_workspace_bytes(int height, int width) {
  if (height <= 0 || width <= 0) return -1;
  const long long ph = (height + 15) / 16 * 16, pw = (width + 15) / 16 * 16;
  // component planes + upsampled row pairs + (multi-scan files) the 16-bit coefficient image
  return 3 * ph * pw + 2 * 3 * (pw + 16) * 2 + 1024 + 3 * ph * pw * 2 + 64;
}

extern "C" int c2d_jpeg_decode_rgb(const uint8_t* data, long long n, uint8_t* out, int height,
                                   int width, void* workspace, long long workspace_bytes) {
  if (!data || !out || !workspace) return C2D_ERR_INVALID_ARG;
  Header hd;
  int rc = parse_header(data, n, hd);
  if (rc) return rc;
  if (hd.height != height || hd.width != width) return C2D_ERR_INVALID_ARG;
  if (workspace_bytes < c2d_jpeg_workspace_bytes(height, width)) return C2D_ERR_WORKSPACE;
  const int mcuw = 8 * hd.maxh, mcuh = 8 * hd.maxv;
  const int mcux = (width + mcuw - 1) / mcuw, mcuy = (height + mcuh - 1) / mcuh;
  uint8_t* ws = (uint8_t*)workspace;
  for (int i = 0; i < hd.ncomp; ++i) {
    Comp& c = hd.comp[i];
    if (hd.single_scan && (!hd.have_q[c.tq] || !hd.dc[c.td].present || !hd.ac[c.ta].present))
      return C2D_ERR_DATA;
    c.dw = (width * c.h + hd.maxh - 1) / hd.maxh;
    c.dh = (height * c.v + hd.maxv - 1) / hd.maxv;
    c.pw = mcux * c.h * 8; c.ph = mcuy * c.v * 8;
    c.plane = ws; ws += (size_t)c.pw * c.ph;
    c.dc_pred = 0;
  }
  if (!hd.single_scan) {
    // ---- multi-scan file: all scans into the coefficient image, then dequantise + IDCT ----
    ws = (uint8_t*)(((uintptr_t)ws + 63) & ~(uintptr_t)63);
    short* cimg[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < hd.ncomp; ++i) {
      cimg[i] = (short*)ws;
      const size_t bytes = (size_t)hd.comp[i].pw * hd.comp[i].ph * 2;
      memset(cimg[i], 0, bytes);
      ws += bytes;
    }
    // (the per-component table checks above looked at the FIRST scan's selectors only when it
    // was interleaved; decode_scans re-checks every scan's tables)
    rc = decode_scans(data, n, hd, cimg);
    if (rc) return rc;
    int blk[64];
    for (int i = 0; i < hd.ncomp; ++i) {
      Comp& c = hd.comp[i];
      if (!hd.have_q[c.tq]) return C2D_ERR_DATA;
      const uint16_t* q = hd.quant[c.tq];
      const int pbw = c.pw / 8, pbh = c.ph / 8;
      for (int by = 0; by < pbh; ++by)
        for (int bx = 0; bx < pbw; ++bx) {
          const short* src = cimg[i] + ((size_t)by * pbw + bx) * 64;
          for (int k = 0; k < 64; ++k) blk[k] = clampc((long long)src[k] * q[k]);
          idct_islow(blk, c.plane + (size_t)by * 8 * c.pw + bx * 8, c.pw);
        }
    }
  } else {
  // ---- entropy decode + dequantize + IDCT, MCU by MCU -----------------------------------
  BitReader br = {hd.scan, data + n, 0, 0, false};
  int coef[64];
  int rst_left = hd.restart, next_rst = 0;
  for (int my = 0; my < mcuy && !br.eof; ++my)
    for (int mx = 0; mx < mcux; ++mx) {
      if (hd.restart && rst_left == 0) {
        // byte-align, expect RSTn
        br.bits = 0; br.buf = 0;
        const uint8_t* q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) ++q;
        if (q + 1 >= br.end || q[1] != 0xd0 + next_rst) return C2D_ERR_DATA;
        br.p = q + 2; br.marker = false; br.pad = 0;
        next_rst = (next_rst + 1) & 7;
        rst_left = hd.restart;
        for (int i = 0; i < hd.ncomp; ++i) hd.comp[i].dc_pred = 0;
      }
      for (int i = 0; i < hd.ncomp; ++i) {
        Comp& c = hd.comp[i];
        const uint16_t* q = hd.quant[c.tq];
        for (int by = 0; by < c.v; ++by)
          for (int bx = 0; bx < c.h; ++bx) {
            memset(coef, 0, sizeof(coef));
            int t = decode_sym(br, hd.dc[c.td]);
            if (t < 0 || t > 11) return C2D_ERR_DATA;
            int diff = t ? extend(get_bits(br, t), t) : 0;
            c.dc_pred += diff;
            if (c.dc_pred > (1 << 20)) c.dc_pred = 1 << 20;       // (only corrupted streams get here)
            if (c.dc_pred < -(1 << 20)) c.dc_pred = -(1 << 20);
            coef[0] = clampc((long long)c.dc_pred * q[0]);
            for (int k = 1; k < 64;) {
              const int rs = decode_sym(br, hd.ac[c.ta]);
              if (rs < 0) return C2D_ERR_DATA;
              const int r = rs >> 4, s = rs & 15;
              if (s == 0) {
                if (r == 15) { k += 16; continue; }
                break;                                        // EOB
              }
              k += r;
              if (k > 63) return C2D_ERR_DATA;
              const int z = kZigzag[k];
              coef[z] = clampc((long long)extend(get_bits(br, s), s) * q[z]);
              ++k;
            }
            uint8_t* dst = c.plane + (size_t)((my * c.v + by) * 8) * c.pw + (mx * c.h + bx) * 8;
            idct_islow(coef, dst, c.pw);
          }
      }
      if (hd.restart) --rst_left;
    }
  if (br.eof) return C2D_ERR_DATA;            // truncated entropy data
  }
  // ---- upsample + colour convert, one output row at a time -------------------------------
  if (hd.ncomp == 1) {
    const Comp& y = hd.comp[0];
    for (int r = 0; r < height; ++r)
      for (int x = 0; x < width; ++x) {
        const uint8_t v = y.plane[(size_t)r * y.pw + x];
        uint8_t* o = out + ((size_t)r * width + x) * 3;
        o[0] = o[1] = o[2] = v;
      }
    return C2D_OK;
  }
  uint8_t* rowbuf = ws;                    // [2 chroma comps][width + 16] upsampled samples
  const int rstride = width + 16;
  for (int r = 0; r < height; ++r) {
    const uint8_t* yrow = hd.comp[0].plane + (size_t)(r * hd.comp[0].v / hd.maxv) * hd.comp[0].pw;
    if (hd.comp[0].h != hd.maxh || hd.comp[0].v != hd.maxv) return C2D_ERR_UNSUPPORTED;
    for (int ci = 1; ci < 3; ++ci) {
      const Comp& c = hd.comp[ci];
      uint8_t* up = rowbuf + (ci - 1) * rstride;
      const int hx = hd.maxh / c.h, vx = hd.maxv / c.v;
      if (hx == 1 && vx == 1) {
        memcpy(up, c.plane + (size_t)r * c.pw, (size_t)width);
      } else if (hx == 2 && vx == 1) {
        // jdsample.c h2v1_fancy_upsample: 3/4 nearer + 1/4 further, biases 1 and 2
        const uint8_t* in = c.plane + (size_t)r * c.pw;
        const int dw = c.dw;
        if (dw == 1) { up[0] = in[0]; up[1] = in[0]; }
        else {
          up[0] = in[0];
          up[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
          for (int x = 1; x < dw - 1; ++x) {
            const int v = in[x] * 3;
            up[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2);
            up[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2);
          }
          up[2 * (dw - 1)] = (uint8_t)((in[dw - 1] * 3 + in[dw - 2] + 1) >> 2);
          up[2 * (dw - 1) + 1] = in[dw - 1];
        }
      } else if (hx == 2 && vx == 2) {
        // jdsample.c h2v2_fancy_upsample: vertical 3/4 + 1/4 into 16-bit sums, then horizontal
        // (3*this + neighbour + 8) >> 4 / (+7) >> 4
        const int cr = r >> 1;
        const int other = (r & 1) ? (cr + 1 < c.dh ? cr + 1 : c.dh - 1) : (cr > 0 ? cr - 1 : 0);
        const uint8_t* in0 = c.plane + (size_t)cr * c.pw;       // nearer row
        const uint8_t* in1 = c.plane + (size_t)other * c.pw;    // further row
        const int dw = c.dw;
        if (dw == 1) {
          const int s = in0[0] * 3 + in1[0];
          up[0] = (uint8_t)((s * 4 + 8) >> 4);
          up[1] = (uint8_t)((s * 4 + 7) >> 4);
        } else {
          int thiscol = in0[0] * 3 + in1[0];
          int nextcol = in0[1] * 3 + in1[1];
          up[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
          up[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
          int lastcol = thiscol; thiscol = nextcol;
          for (int x = 1; x < dw - 1; ++x) {
            nextcol = in0[x + 1] * 3 + in1[x + 1];
            up[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
            up[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol; thiscol = nextcol;
          }
          up[2 * (dw - 1)] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
          up[2 * (dw - 1) + 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
        }
      } else if (hx == 1 && vx == 2) {
        // jdsample.c h1v2_fancy_upsample (libjpeg-turbo): 3/4 nearer + 1/4 further row,
        // bias 1 for the upper output row and 2 for the lower one
        const int cr = r >> 1;
        const int other = (r & 1) ? (cr + 1 < c.dh ? cr + 1 : c.dh - 1) : (cr > 0 ? cr - 1 : 0);
        const uint8_t* in0 = c.plane + (size_t)cr * c.pw;
        const uint8_t* in1 = c.plane + (size_t)other * c.pw;
        const int bias = (r & 1) ? 2 : 1;
        for (int x = 0; x < width; ++x) up[x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
      } else {
        return C2D_ERR_UNSUPPORTED;
      }
    }
    // jdcolor.c ycc_rgb_convert: SCALEBITS = 16 fixed point
    const uint8_t* cb = rowbuf;
    const uint8_t* cr_ = rowbuf + rstride;
    uint8_t* o = out + (size_t)r * width * 3;
    for (int x = 0; x < width; ++x) {
      const int y = yrow[x], b = cb[x] - 128, rr = cr_[x] - 128;
      const int crr = (91881 * rr + 32768) >> 16;                         // FIX(1.40200)
      const int cbb = (116130 * b + 32768) >> 16;                         // FIX(1.77200)
      const int g = (-22554 * b + (-46802 * rr + 32768)) >> 16;           // FIX(0.34414), FIX(0.71414)
      o[3 * x] = clamp8(y + crr);
      o[3 * x + 1] = clamp8(y + g);
      o[3 * x + 2] = clamp8(y + cbb);
    }
  }
  return C2D_OK;
}
