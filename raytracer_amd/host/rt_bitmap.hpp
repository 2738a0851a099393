// rt_bitmap.hpp -- Color / Bitmap and their free functions with the reference's names
// (raytracer_cpp/writebitmap.h:5-182).  pData is the row-major, top-down RGB8 array that
// rt_download_framebuffer(RT_FMT_RGB8) fills; WriteBitmap emits the same bytes as the reference's
// writer through rt_write_bmp (row-buffered instead of one fprintf per byte).
#pragma once
#include <cassert>
#include <cstdint>
#include <cstdlib>

#include "rt_api.h"

struct Color {
    uint8_t red, green, blue;
};

struct Bitmap {
    int width;
    int height;
    Color *pData;
};

inline Bitmap *CreateBitmap(int width, int height)
{
    Bitmap *b = static_cast<Bitmap *>(std::malloc(sizeof(Bitmap)));
    b->width = width;
    b->height = height;
    b->pData = static_cast<Color *>(std::malloc(sizeof(Color) * (size_t)width * (size_t)height));
    return b;
}

inline void DestroyBitmap(Bitmap *b)
{
    if (!b) return;
    std::free(b->pData);
    std::free(b);
}

inline Color &GetPixel(Bitmap *b, int x, int y)
{
    static Color outside{};
    if (x < 0 || y < 0 || x >= b->width || y >= b->height) {
        assert(!"GetPixel out of bounds");
        return outside;
    }
    return b->pData[(size_t)b->width * (size_t)y + (size_t)x];
}

inline void PutPixel(Bitmap *b, int x, int y, const Color &c)
{
#ifdef DEBUG
    if (x < 0 || y < 0 || x >= b->width || y >= b->height) {
        assert(!"PutPixel out of bounds");
        return;
    }
#endif
    b->pData[(size_t)b->width * (size_t)y + (size_t)x] = c;
}

inline void ClearBitmap(Bitmap *b, const Color &c)
{
    for (size_t i = 0, n = (size_t)b->width * (size_t)b->height; i < n; i++) b->pData[i] = c;
}

inline void WriteBitmap(Bitmap *b, const char *filename)
{
    static_assert(sizeof(Color) == 3, "Color must be packed RGB8");
    rt_write_bmp(filename, b->width, b->height, reinterpret_cast<const uint8_t *>(b->pData));
}
