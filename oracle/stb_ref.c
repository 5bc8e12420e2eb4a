/* stb_ref.c — TEST INFRASTRUCTURE: a reference build of the image decoder the reference itself uses.
 *
 * Nexus decodes every texture and glTF image with stbi_load / stbi_load_from_memory asking for four channels
 * (/root/reference/Nexus/src/Assets/IMGLoader.cpp:17-41); stb_image.h is vendored with it
 * (/root/reference/Nexus/vendor/stb/stb_image.h, a single C header with no dependencies), so this part of the reference
 * compiles here as it stands.  `make -C oracle ref` compiles THIS file against the header where it lies under
 * /root/reference (nothing of it is copied into the repository) into oracle/_ref/libstbref.so, which the tests load to check
 * the product's own decoders (nexus_amd/csrc/host/IMGLoader.cpp, JPEGDecoder.cpp) byte for byte, and which
 * tests/golden/make_image_golden.py uses to write the committed expected outputs.  Never linked into the product. */
#include <stdlib.h>
#include <string.h>

/* stb_image's documented allocator hooks, pointed at a zero-filling allocator: on a DAMAGED file the library may return pixels
 * it never wrote (component planes and coefficient blocks come from malloc), i.e. whatever the heap held.  Zero-filled they are
 * reproducible, and equal to what the product's decoder (which zero-initialises the same buffers) leaves there; on intact files
 * the hooks change nothing. */
#define STBI_MALLOC(sz) calloc(1, sz)
#define STBI_REALLOC(p, newsz) realloc(p, newsz)
#define STBI_FREE(p) free(p)
/* STBI_NO_SIMD: the library's portable code paths.  Its SSE2 kernels are, by its own account, bit-identical to them — on
 * sample values a well-formed file can produce; on the out-of-range coefficients of a damaged file the 16-bit SIMD lanes
 * saturate where the portable code wraps.  The portable paths are the ones the product's decoder restates. */
#define STBI_NO_SIMD
#define STB_IMAGE_IMPLEMENTATION
#define STBI_NO_STDIO
#include "stb_image.h"

/* RGBA8 of an image file in memory, as Assets/IMGLoader.cpp asks for it.  Returns 0 and the size, or -1 (not decodable). */
int nxref_image_size(const unsigned char *data, int len, int *w, int *h, int *channels)
{
    unsigned char *px = stbi_load_from_memory(data, len, w, h, channels, 4);
    if (!px) return -1;
    stbi_image_free(px);
    return 0;
}

int nxref_image_decode(const unsigned char *data, int len, unsigned char *dst, size_t capacity)
{
    int w = 0, h = 0, c = 0;
    unsigned char *px = stbi_load_from_memory(data, len, &w, &h, &c, 4);
    if (!px) return -1;
    if ((size_t)w * (size_t)h * 4 > capacity) {
        stbi_image_free(px);
        return -2;
    }
    memcpy(dst, px, (size_t)w * (size_t)h * 4);
    stbi_image_free(px);
    return 0;
}

const char *nxref_failure_reason(void) { return stbi_failure_reason(); }
