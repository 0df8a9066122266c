/*
 * jpeg.c -- baseline JPEG (JFIF) writer for 8-bit RGB images.
 *
 * The reference writes .jpg/.jpeg at quality 90 through a vendored third-party
 * encoder (c/main.c:80-81, 430-435).  This one is the build's own, written from
 * the standard (ITU-T T.81): sequential DCT, Huffman coding, three components
 * Y Cb Cr without subsampling, the example quantisation tables of Annex K.1
 * scaled for the quality, the example Huffman tables of Annex K.3.  It does not
 * produce the reference encoder's bytes (no two encoders do); what it produces
 * is a conforming file of the same picture -- the tests decode it with an
 * independent decoder and compare the pixels with the source.
 */
#include "rt_host.h"

#include <math.h>
#include <string.h>

/* zig-zag scan order: position k of the coded sequence -> index 8*row + col (T.81 Figure A.6) */
static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* T.81 Table K.1 (luminance) and K.2 (chrominance), in row order */
static const uint8_t kQuantLuma[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                       14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                       18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                       49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t kQuantChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                         99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                         99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

/* T.81 Tables K.3 - K.6: number of codes of each length 1..16, then the symbols in code order */
static const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

typedef struct {
  uint16_t code[256];
  uint8_t len[256];
} huff_table;

/* T.81 Annex C: codes of increasing length in symbol order, starting from zero */
static void huff_build(huff_table *t, const uint8_t bits[16], const uint8_t *vals) {
  memset(t, 0, sizeof *t);
  unsigned code = 0, k = 0;
  for (unsigned len = 1; len <= 16; ++len) {
    for (unsigned i = 0; i < bits[len - 1]; ++i, ++k) {
      t->code[vals[k]] = (uint16_t)code++;
      t->len[vals[k]] = (uint8_t)len;
    }
    code <<= 1;
  }
}

typedef struct {
  FILE *fp;
  uint32_t acc;   /* bits not yet written, right-aligned */
  int nacc;
  int failed;
} bit_writer;

static void put_byte(bit_writer *w, unsigned b) {
  if (fputc((int)(b & 0xff), w->fp) == EOF) w->failed = 1;
}

static void put_bits(bit_writer *w, unsigned value, int n) {
  w->acc = (w->acc << n) | (value & ((1u << n) - 1u));
  w->nacc += n;
  while (w->nacc >= 8) {
    const unsigned b = (w->acc >> (w->nacc - 8)) & 0xffu;
    put_byte(w, b);
    if (b == 0xffu) put_byte(w, 0);   /* byte stuffing, T.81 B.1.1.5 */
    w->nacc -= 8;
  }
}

static void put_marker(bit_writer *w, unsigned marker, unsigned payload_len) {
  put_byte(w, 0xff);
  put_byte(w, marker);
  put_byte(w, (payload_len + 2) >> 8);
  put_byte(w, (payload_len + 2) & 0xff);
}

/* magnitude category and the extra bits of a coefficient (T.81 F.1.2.1, F.1.2.2) */
static int category(int v, unsigned *extra) {
  int a = v < 0 ? -v : v, n = 0;
  while (a) { ++n; a >>= 1; }
  *extra = (unsigned)(v < 0 ? v - 1 : v) & ((1u << n) - 1u);
  return n;
}

/* one 8x8 block of samples (level shift already applied) -> quantised coefficients in zig-zag order */
static void transform_block(const float in[64], const float basis[8][8], const float recip[64], int out[64]) {
  float tmp[64], freq[64];
  for (int y = 0; y < 8; ++y)          /* rows */
    for (int u = 0; u < 8; ++u) {
      float s = 0;
      for (int x = 0; x < 8; ++x) s += in[8 * y + x] * basis[u][x];
      tmp[8 * y + u] = s;
    }
  for (int u = 0; u < 8; ++u)          /* columns */
    for (int v = 0; v < 8; ++v) {
      float s = 0;
      for (int y = 0; y < 8; ++y) s += tmp[8 * y + u] * basis[v][y];
      freq[8 * v + u] = s;
    }
  for (int k = 0; k < 64; ++k) {
    const float q = freq[kZigzag[k]] * recip[kZigzag[k]];
    out[k] = (int)(q < 0 ? q - 0.5f : q + 0.5f);
  }
}

static void code_block(bit_writer *w, const int zz[64], int *dc_pred, const huff_table *dc, const huff_table *ac) {
  unsigned extra;
  const int diff = zz[0] - *dc_pred;
  *dc_pred = zz[0];
  int n = category(diff, &extra);
  put_bits(w, dc->code[n], dc->len[n]);
  if (n) put_bits(w, extra, n);
  int last = 63;
  while (last > 0 && zz[last] == 0) --last;
  int run = 0;
  for (int k = 1; k <= last; ++k) {
    if (zz[k] == 0) { ++run; continue; }
    while (run > 15) { put_bits(w, ac->code[0xf0], ac->len[0xf0]); run -= 16; }   /* ZRL */
    n = category(zz[k], &extra);
    put_bits(w, ac->code[(run << 4) | n], ac->len[(run << 4) | n]);
    put_bits(w, extra, n);
    run = 0;
  }
  if (last < 63) put_bits(w, ac->code[0x00], ac->len[0x00]);                     /* EOB */
}

int rth_write_jpg(FILE *fp, const uint8_t *rgb8, size_t w, size_t h, int quality) {
  if (!fp || !rgb8 || w == 0 || h == 0 || w > 65535 || h > 65535) { rth_set_error("bad image for the JPEG writer"); return RT_ERR_ARG; }
  if (quality < 1) quality = 1;
  if (quality > 100) quality = 100;
  /* the usual scaling of the example tables: 50 leaves them as they are, 100 makes them all ones */
  const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
  uint8_t qt[2][64];
  for (int k = 0; k < 64; ++k) {
    int l = ((int)kQuantLuma[k] * scale + 50) / 100, c = ((int)kQuantChroma[k] * scale + 50) / 100;
    qt[0][k] = (uint8_t)(l < 1 ? 1 : l > 255 ? 255 : l);
    qt[1][k] = (uint8_t)(c < 1 ? 1 : c > 255 ? 255 : c);
  }
  /* orthonormal DCT-II basis: F(u) = sum_x basis[u][x] f(x); two passes give T.81 A.3.3 including its 1/4 C(u) C(v) */
  float basis[8][8], recip[2][64];
  for (int u = 0; u < 8; ++u)
    for (int x = 0; x < 8; ++x)
      basis[u][x] = (float)((u == 0 ? sqrt(0.125) : 0.5) * cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0));
  for (int k = 0; k < 64; ++k) { recip[0][k] = 1.0f / (float)qt[0][k]; recip[1][k] = 1.0f / (float)qt[1][k]; }
  huff_table dc[2], ac[2];
  huff_build(&dc[0], kDcLumaBits, kDcVals);
  huff_build(&dc[1], kDcChromaBits, kDcVals);
  huff_build(&ac[0], kAcLumaBits, kAcLumaVals);
  huff_build(&ac[1], kAcChromaBits, kAcChromaVals);

  bit_writer bw = {fp, 0, 0, 0};
  put_byte(&bw, 0xff); put_byte(&bw, 0xd8);                                     /* SOI */
  put_marker(&bw, 0xe0, 14);                                                    /* APP0: JFIF 1.01, no density, no thumbnail */
  { const uint8_t jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0}; for (int i = 0; i < 14; ++i) put_byte(&bw, jfif[i]); }
  for (int t = 0; t < 2; ++t) {                                                 /* DQT, 8-bit entries in zig-zag order */
    put_marker(&bw, 0xdb, 65);
    put_byte(&bw, (unsigned)t);
    for (int k = 0; k < 64; ++k) put_byte(&bw, qt[t][kZigzag[k]]);
  }
  put_marker(&bw, 0xc0, 15);                                                    /* SOF0: baseline, 8 bits, 3 components, 1x1 sampling */
  put_byte(&bw, 8);
  put_byte(&bw, (unsigned)(h >> 8)); put_byte(&bw, (unsigned)(h & 0xff));
  put_byte(&bw, (unsigned)(w >> 8)); put_byte(&bw, (unsigned)(w & 0xff));
  put_byte(&bw, 3);
  for (unsigned c = 0; c < 3; ++c) { put_byte(&bw, c + 1); put_byte(&bw, 0x11); put_byte(&bw, c ? 1 : 0); }
  {                                                                             /* DHT: class (0 DC, 1 AC) << 4 | id */
    const uint8_t *bits[4] = {kDcLumaBits, kAcLumaBits, kDcChromaBits, kAcChromaBits};
    const uint8_t *vals[4] = {kDcVals, kAcLumaVals, kDcVals, kAcChromaVals};
    const unsigned ids[4] = {0x00, 0x10, 0x01, 0x11};
    for (int t = 0; t < 4; ++t) {
      unsigned n = 0;
      for (int i = 0; i < 16; ++i) n += bits[t][i];
      put_marker(&bw, 0xc4, 17 + n);
      put_byte(&bw, ids[t]);
      for (int i = 0; i < 16; ++i) put_byte(&bw, bits[t][i]);
      for (unsigned i = 0; i < n; ++i) put_byte(&bw, vals[t][i]);
    }
  }
  put_marker(&bw, 0xda, 10);                                                    /* SOS: three components, whole spectrum */
  put_byte(&bw, 3);
  for (unsigned c = 0; c < 3; ++c) { put_byte(&bw, c + 1); put_byte(&bw, c ? 0x11 : 0x00); }
  put_byte(&bw, 0); put_byte(&bw, 63); put_byte(&bw, 0);

  int pred[3] = {0, 0, 0};
  for (size_t by = 0; by < h; by += 8)
    for (size_t bx = 0; bx < w; bx += 8) {
      float comp[3][64];
      for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
          const size_t sy = by + (size_t)y < h ? by + (size_t)y : h - 1, sx = bx + (size_t)x < w ? bx + (size_t)x : w - 1;   /* edge blocks repeat the border */
          const uint8_t *p = rgb8 + 3 * (sy * w + sx);
          const float r = p[0], g = p[1], b = p[2];
          comp[0][8 * y + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;                /* JFIF: full-range BT.601 */
          comp[1][8 * y + x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
          comp[2][8 * y + x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
        }
      for (int c = 0; c < 3; ++c) {
        int zz[64];
        transform_block(comp[c], basis, recip[c ? 1 : 0], zz);
        code_block(&bw, zz, &pred[c], &dc[c ? 1 : 0], &ac[c ? 1 : 0]);
      }
    }
  if (bw.nacc) put_bits(&bw, 0x7f, 8 - bw.nacc);                                /* pad the last byte with ones */
  put_byte(&bw, 0xff); put_byte(&bw, 0xd9);                                     /* EOI */
  if (bw.failed) { rth_set_error("could not write the JPEG file"); return RT_ERR_ARG; }
  return RT_OK;
}
