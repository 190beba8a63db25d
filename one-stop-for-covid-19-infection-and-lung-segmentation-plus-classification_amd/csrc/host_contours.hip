// Host-side (CPU) contour search of the lung cropper: cv2.findContours(img, RETR_TREE, CHAIN_APPROX_SIMPLE) + cv2.contourArea + cv2.boundingRect of
// every contour of a batch of uint8 slices (task1_preprocessing_plus_unet_with_comments.py:219-233, task3_lung_segmentation_unet.py:221-235).
//
// Border following (Suzuki & Abe 1985, the algorithm OpenCV's contours.cpp implements) is a serial walk per border, so it stays on the host like the
// reference's OpenCV call -- native and threaded over slices instead of one Python call per slice.  Only what `cropper` consumes is produced: per contour
// the polygon area and the bounding rectangle, in cv2's output order; the point lists are never materialised (the shoelace sum and the min / max are
// accumulated while walking).  No device code in this file.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <thread>
#include <vector>

#include "common.h"

namespace {

const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

struct Border {
  long long shoelace;        // 2 x signed area of the pixel-centre polygon
  int x0, y0, x1, y1;        // bounding box, framed coordinates
  int is_hole, parent, label;
};

// follow one border of the framed label image f (row pitch `ld`) from (x, y); marks the pixels it passes with +-nbd
void follow(int* f, int ld, int x, int y, int is_hole, int nbd, Border& b) {
  int s_end = is_hole ? 0 : 4, s = s_end;
  do {                                                        // first neighbour: clockwise from the background side
    s = (s - 1) & 7;
    if (f[(y + DY[s]) * ld + x + DX[s]] != 0) break;
  } while (s != s_end);
  b.shoelace = 0; b.x0 = b.x1 = x; b.y0 = b.y1 = y;
  if (f[(y + DY[s]) * ld + x + DX[s]] == 0) {                 // single-pixel component
    f[y * ld + x] = -nbd;
    return;
  }
  const int x1 = x + DX[s], y1 = y + DY[s];
  int x3 = x, y3 = y;
  for (;;) {
    const int se = s;
    int x4, y4;
    for (;;) {                                                // next neighbour: counter-clockwise from the one after the previous point
      ++s;
      x4 = x3 + DX[s & 7]; y4 = y3 + DY[s & 7];
      if (f[y4 * ld + x4] != 0) break;
    }
    s &= 7;
    int& cur = f[y3 * ld + x3];
    if ((unsigned)(s - 1) < (unsigned)se) cur = -nbd;         // the right neighbour was examined and is empty
    else if (cur == 1) cur = nbd;
    b.shoelace += (long long)x3 * y4 - (long long)y3 * x4;
    b.x0 = std::min(b.x0, x4); b.x1 = std::max(b.x1, x4); b.y0 = std::min(b.y0, y4); b.y1 = std::max(b.y1, y4);
    if (x4 == x && y4 == y && x3 == x1 && y3 == y1) break;
    x3 = x4; y3 = y4;
    s = (s + 4) & 7;
  }
}

// -> number of contours of the image (all of them are counted; the first `cap` are written, in cv2's order)
int contours_of(const uint8_t* img, int h, int w, int cap, double* areas, int32_t* rects, std::vector<int>& f, std::vector<Border>& found) {
  const int ld = w + 2;
  f.assign((size_t)(h + 2) * ld, 0);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) f[(size_t)(y + 1) * ld + x + 1] = img[(size_t)y * w + x] != 0;
  found.clear();
  std::vector<int> hole_of(2, 1), parent_of(2, 1);            // by label; label 1 = the frame, which counts as a hole border
  int nbd = 1;
  for (int y = 1; y <= h; ++y) {
    int* row = f.data() + (size_t)y * ld;
    int lnbd = 1, prev = 0;
    for (int x = 1; x <= w + 1; ++x) {
      int p = row[x];
      if (p != prev) {
        int sx = -1, is_hole = 0;
        if (prev == 0 && p == 1) sx = x;
        else if (p == 0 && prev >= 1) {
          if (prev > 1) lnbd = prev;
          sx = x - 1; is_hole = 1;
        }
        if (sx >= 0) {
          ++nbd;
          const int parent = hole_of[lnbd] == is_hole ? parent_of[lnbd] : lnbd;      // Suzuki's table 1
          hole_of.push_back(is_hole); parent_of.push_back(parent);
          Border b;
          follow(f.data(), ld, sx, y, is_hole, nbd, b);
          b.is_hole = is_hole; b.parent = parent; b.label = nbd;
          found.push_back(b);
          p = row[x];
        }
        prev = p;
        if (p != 0 && p != 1) lnbd = std::abs(p);
      }
    }
  }
  // cv2's order: the tree in pre-order, siblings in reverse order of discovery (a new contour is pushed at the head of its parent's child list)
  const int nc = (int)found.size();
  std::vector<int> first_child(nbd + 1, -1), next_sib(nc, -1);
  for (int i = 0; i < nc; ++i) {                              // pushing in discovery order at the head = reversed list
    next_sib[i] = first_child[found[i].parent];
    first_child[found[i].parent] = i;
  }
  std::vector<int> stack;
  int written = 0;
  for (int i = first_child[1]; i >= 0;) {
    if (written < cap) {
      const Border& b = found[i];
      areas[written] = 0.5 * (double)(b.shoelace < 0 ? -b.shoelace : b.shoelace);
      rects[4 * written + 0] = b.x0 - 1; rects[4 * written + 1] = b.y0 - 1; rects[4 * written + 2] = b.x1 - b.x0 + 1; rects[4 * written + 3] = b.y1 - b.y0 + 1;
    }
    ++written;
    const int child = first_child[found[i].label];
    if (child >= 0) {
      stack.push_back(i);
      i = child;
      continue;
    }
    i = next_sib[i];
    while (i < 0 && !stack.empty()) {
      i = next_sib[stack.back()];
      stack.pop_back();
    }
  }
  return written;
}

}  // namespace

extern "C" int32_t unet_pre_contours_u8(unet_ctx* ctx, const uint8_t* imgs, int32_t n, int32_t h, int32_t w, int32_t max_contours, double* areas,
                                        int32_t* rects, int32_t* counts, int32_t threads) {
  if (!imgs || !areas || !rects || !counts || n < 1 || h < 1 || w < 1 || max_contours < 1) UNET_FAIL(ctx, UNET_E_ARG, "pre_contours_u8: bad args");
  if ((long long)(h + 2) * (w + 2) > (1ll << 30)) UNET_FAIL(ctx, UNET_E_SHAPE, "pre_contours_u8: image %dx%d too large", h, w);
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  nt = std::max(1, std::min(nt, std::min((int)n, 64)));
  std::atomic<int> next(0);
  auto work = [&]() {
    std::vector<int> f;
    std::vector<Border> found;
    for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1))
      counts[i] = contours_of(imgs + (size_t)i * h * w, h, w, max_contours, areas + (size_t)i * max_contours, rects + (size_t)i * max_contours * 4, f, found);
  };
  if (nt == 1) {
    work();
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back(work);
    for (auto& t : pool) t.join();
  }
  return UNET_OK;
}
