/* tau_4splat.h — writer for the `.4spl` voxel-video container th3cs.cu exports and viewer.html plays.
 *
 * The reference links `4splat.c` for this (th3cs.cu:2, API declared at th3cs.cu:21-63) but does not ship it, so
 * this is a from-scratch writer with the same type and function names.  What pins the layout is the one reader the
 * reference does ship, viewer.html:67-96 (all little-endian):
 *   offset 0    header, 32 bytes: magic u32 | version u8[4] | width | height | depth | frames | pSize | flags (u32 each)
 *   offset 32   palette, pSize x 48 bytes: 12 fp32 per entry (mu_x sigma_x mu_y sigma_y mu_z sigma_z mu_t sigma_t r g b alpha)
 *   then        indices, frames x depth x height x width entries, x fastest ((z*height + y)*width + x, viewer.html:128),
 *               one byte each when flags == 0x0004 ("Float32 precision (0x04), 8-bit index width (0x00)", th3cs.cu:1226)
 *   then        footer: checksum u32 | idxoffset u64 | end u32 (packed, 16 bytes)
 * The viewer reads width..pSize, the palette colours and the indices; it never looks at magic, version, flags or
 * the footer, and 4splat.c is not there to say what they hold.  This writer's choices for those words:
 *   magic 0x4C505334 (the bytes "4SPL"), version {1,0,0,0}, flags bits 0-1 = log2(bytes per index), bit 2 = fp32
 *   palette; checksum = CRC-32 (IEEE 802.3) of every byte before the footer, idxoffset = 32 + 48 pSize, end
 *   0x444E4534 (the bytes "4END"). */
#ifndef TAU_4SPLAT_H
#define TAU_4SPLAT_H
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef struct { float mu_x, sigma_x, mu_y, sigma_y, mu_z, sigma_z, mu_t, sigma_t, r, g, b, alpha; } Splat4D;
typedef struct { uint32_t magic; uint8_t version[4]; uint32_t width, height, depth, frames; uint32_t pSize; uint32_t flags; } Splat4DHeader;
typedef struct { Splat4D *palette; } Splat4DPalette;
typedef struct { uint64_t *index; } Splat4DIndex;
typedef struct { uint32_t checksum; uint64_t idxoffset; uint32_t end; } Splat4DFooter;
typedef struct { Splat4DHeader header; Splat4DPalette palette; Splat4DIndex index; Splat4DFooter footer; } Splat4DVideo;

#define SPLAT4D_MAGIC 0x4C505334u
#define SPLAT4D_END 0x444E4534u

static inline Splat4D create_splat4D(float mu_x, float sigma_x, float mu_y, float sigma_y, float mu_z, float sigma_z,
                                     float mu_t, float sigma_t, float r, float g, float b, float alpha) {
  Splat4D s = {mu_x, sigma_x, mu_y, sigma_y, mu_z, sigma_z, mu_t, sigma_t, r, g, b, alpha};
  return s;
}
static inline Splat4DHeader create_splat4DHeader(uint32_t width, uint32_t height, uint32_t depth, uint32_t frames,
                                                 uint32_t pSize, uint32_t flags) {
  Splat4DHeader h;
  h.magic = SPLAT4D_MAGIC;
  h.version[0] = 1; h.version[1] = 0; h.version[2] = 0; h.version[3] = 0;
  h.width = width; h.height = height; h.depth = depth; h.frames = frames; h.pSize = pSize; h.flags = flags;
  return h;
}
static inline Splat4DVideo create_splat4DVideo(Splat4DHeader header, Splat4D *splats, uint64_t *idxs) {
  Splat4DVideo v;
  v.header = header;
  v.palette.palette = splats;
  v.index.index = idxs;
  v.footer.checksum = 0;
  v.footer.idxoffset = 32u + 48u * (uint64_t)header.pSize;
  v.footer.end = SPLAT4D_END;
  return v;
}

/* ---- byte-order-independent output with a running CRC-32 */
typedef struct { FILE *fp; uint32_t crc; bool ok; } splat4d_out;
static inline uint32_t splat4d_crc_update(uint32_t crc, const uint8_t *p, size_t n) {
  static uint32_t table[256];
  static int have = 0;
  if (!have) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
      table[i] = c;
    }
    have = 1;
  }
  for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
  return crc;
}
static inline void splat4d_put(splat4d_out *o, const void *p, size_t n) {
  if (!o->ok) return;
  o->crc = splat4d_crc_update(o->crc, (const uint8_t *)p, n);
  if (fwrite(p, 1, n, o->fp) != n) o->ok = false;
}
static inline void splat4d_put_u32(splat4d_out *o, uint32_t v) {
  uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
  splat4d_put(o, b, 4);
}
static inline void splat4d_put_f32(splat4d_out *o, float f) {
  uint32_t v;
  memcpy(&v, &f, 4);
  splat4d_put_u32(o, v);
}
static inline bool splat4d_write_head(splat4d_out *o, const Splat4DHeader *h, const Splat4D *palette) {
  splat4d_put_u32(o, h->magic);
  splat4d_put(o, h->version, 4);
  splat4d_put_u32(o, h->width); splat4d_put_u32(o, h->height); splat4d_put_u32(o, h->depth);
  splat4d_put_u32(o, h->frames); splat4d_put_u32(o, h->pSize); splat4d_put_u32(o, h->flags);
  for (uint32_t i = 0; i < h->pSize; i++) {
    const float *f = &palette[i].mu_x;
    for (int k = 0; k < 12; k++) splat4d_put_f32(o, f[k]);
  }
  return o->ok;
}
static inline bool splat4d_write_foot(splat4d_out *o, uint32_t pSize) {
  const uint32_t checksum = o->crc ^ 0xFFFFFFFFu;
  const uint64_t idxoffset = 32u + 48u * (uint64_t)pSize;
  splat4d_put_u32(o, checksum);
  splat4d_put_u32(o, (uint32_t)idxoffset);
  splat4d_put_u32(o, (uint32_t)(idxoffset >> 32));
  splat4d_put_u32(o, SPLAT4D_END);
  return o->ok;
}

/* the reference's entry point: indices held as one uint64_t per voxel, narrowed to the width the flags name */
static inline bool write_splat4DVideo(FILE *fp, Splat4DVideo *v) {
  splat4d_out o = {fp, 0xFFFFFFFFu, true};
  const Splat4DHeader *h = &v->header;
  if (!splat4d_write_head(&o, h, v->palette.palette)) return false;
  const uint64_t n = (uint64_t)h->width * h->height * h->depth * h->frames;
  const unsigned wbytes = 1u << (h->flags & 3u);
  uint8_t buf[4096];
  size_t fill = 0;
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t x = v->index.index[i];
    for (unsigned k = 0; k < wbytes; k++) buf[fill++] = (uint8_t)(x >> (8 * k));
    if (fill + 8 > sizeof(buf)) { splat4d_put(&o, buf, fill); fill = 0; }
  }
  splat4d_put(&o, buf, fill);
  v->footer.checksum = o.crc ^ 0xFFFFFFFFu;
  return splat4d_write_foot(&o, h->pSize);
}

/* the same file from byte indices (flags 0x0004), as the engine hands them back: no 8-byte-per-voxel staging */
static inline bool write_splat4D_u8(FILE *fp, const Splat4DHeader *h, const Splat4D *palette, const uint8_t *idx) {
  if ((h->flags & 3u) != 0u) return false;
  splat4d_out o = {fp, 0xFFFFFFFFu, true};
  if (!splat4d_write_head(&o, h, palette)) return false;
  splat4d_put(&o, idx, (size_t)h->width * h->height * h->depth * h->frames);
  return splat4d_write_foot(&o, h->pSize);
}
#endif
