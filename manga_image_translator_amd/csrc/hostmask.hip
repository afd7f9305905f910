// hostmask.hip — host-side (CPU, no kernels) mask merging of the ctd detector's refine_mask: the per-text-line greedy
// component merge of manga_translator/detection/ctd_utils/textmask.py:74-132 (merge_mask_list, filter_with_lines False,
// pred_thresh 30), which the reference runs on cv2.connectedComponentsWithStats / erode / dilate / bitwise_xor.
//
// Why native: a text-line window holds thousands of candidate components (every glyph fragment of every candidate mask) and
// each is tried against the running merged mask — ~10^5 tiny array operations per page in numpy (2.4 s / page measured), a few
// milliseconds here.  Semantics are exactly those of hostglue._merge_mask_list (its numpy form stays as the test reference):
//   * labels are numbered in raster order of each component's first pixel (scipy.ndimage.label's order; 8-connectivity),
//   * "try_merge": a component joins when  sum(xor(merged | comp, pred)) < sum(xor(merged, pred))  over the component's bounding
//     box — with 0/255 images that is  #(comp & ~merged & pred) > #(comp & ~merged & ~pred),
//   * candidates are visited in ascending xor score (stable), components in label order, bounding boxes of w*h < 3 skipped,
//   * optional 5x5 dilation (refine_mode == REFINEMASK_INPAINT), then hole filling: components of the complement smaller than
//     the second-largest complement area are tried the same way.
// PARITY UNPINNED against the real OpenCV (its component numbering is not raster order for every shape).
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <numeric>
#include <vector>
#include "../../include/mit_hip.h"
#include "common.h"

namespace {

struct Comp {
    int x0, y0, x1, y1;  // inclusive bounding box
    int area;
    int first;           // index into the pixel list
};

// 8-connected components of (img != 0), labels 1.. in raster order of the first pixel.  lab gets 0 for background.
// pixels: all foreground pixel indices grouped by label (ascending raster order inside a label).
int label8(const uint8_t *img, int h, int w, std::vector<int> &lab, std::vector<Comp> &comps, std::vector<int> &pixels) {
    const int n = h * w;
    lab.assign(n, 0);
    std::vector<int> parent(1, 0);
    auto find = [&](int a) {
        while (parent[a] != a) {
            parent[a] = parent[parent[a]];
            a = parent[a];
        }
        return a;
    };
    auto unite = [&](int a, int b) {
        a = find(a);
        b = find(b);
        if (a != b) parent[a > b ? a : b] = a < b ? a : b;  // the smaller provisional label (earlier first pixel) stays root
    };
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            const int i = y * w + x;
            if (!img[i]) continue;
            int l = 0;
            const int nb[4] = {x > 0 ? lab[i - 1] : 0, (y > 0 && x > 0) ? lab[i - w - 1] : 0, y > 0 ? lab[i - w] : 0,
                               (y > 0 && x + 1 < w) ? lab[i - w + 1] : 0};
            for (int k = 0; k < 4; ++k) {
                if (!nb[k]) continue;
                if (!l) l = nb[k];
                else unite(l, nb[k]);
            }
            if (!l) {
                l = (int)parent.size();
                parent.push_back(l);
            }
            lab[i] = l;
        }
    }
    // final labels in raster order of first appearance (a root's first pixel is the component's first pixel, because roots are
    // always the smallest provisional label of their set and provisional labels are created in raster order)
    std::vector<int> final_of(parent.size(), 0);
    int ncomp = 0;
    for (int i = 0; i < n; ++i) {
        if (!lab[i]) continue;
        const int r = find(lab[i]);
        if (!final_of[r]) final_of[r] = ++ncomp;
        lab[i] = final_of[r];
    }
    comps.assign(ncomp + 1, Comp{w, h, -1, -1, 0, 0});
    for (int i = 0; i < n; ++i) {
        const int l = lab[i];
        if (!l) continue;
        Comp &c = comps[l];
        const int y = i / w, x = i - y * w;
        c.x0 = std::min(c.x0, x), c.x1 = std::max(c.x1, x), c.y0 = std::min(c.y0, y), c.y1 = std::max(c.y1, y);
        ++c.area;
    }
    int run = 0;
    for (int l = 1; l <= ncomp; ++l) {
        comps[l].first = run;
        run += comps[l].area;
    }
    pixels.assign(run, 0);
    std::vector<int> fill(ncomp + 1, 0);
    for (int i = 0; i < n; ++i) {
        const int l = lab[i];
        if (l) pixels[comps[l].first + fill[l]++] = i;
    }
    return ncomp;
}

// the pixels of the component, against merged / pred (0 / 255 images): join when it lowers the xor sum
inline void try_merge(const int *px, int count, const uint8_t *pred, uint8_t *merged) {
    int gain = 0, loss = 0;
    for (int k = 0; k < count; ++k) {
        const int i = px[k];
        if (merged[i]) continue;
        if (pred[i]) ++gain;
        else ++loss;
    }
    if (gain > loss)
        for (int k = 0; k < count; ++k) merged[px[k]] = 255;
}

}  // namespace

extern "C" int mit_merge_mask_list(const uint8_t *cands, const int64_t *scores, int n_cands, const uint8_t *pred_mask, int h, int w,
                                   int inpaint_dilate, uint8_t *merged) {
    if (!cands || !scores || !pred_mask || !merged) return mit_set_error("mit_merge_mask_list: null pointer");
    if (n_cands <= 0 || h <= 0 || w <= 0 || (int64_t)h * w > 0x3fffffffLL) return mit_set_error("mit_merge_mask_list: bad size");
    const int n = h * w;
    // pred: 3x3 cross erosion with +inf outside (cv2.erode), then > 60 -> 255 (textmask.py:77-81, pred_thresh 30 * 2)
    std::vector<uint8_t> pred(n);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int i = y * w + x;
            uint8_t m = pred_mask[i];
            if (x > 0) m = std::min(m, pred_mask[i - 1]);
            if (x + 1 < w) m = std::min(m, pred_mask[i + 1]);
            if (y > 0) m = std::min(m, pred_mask[i - w]);
            if (y + 1 < h) m = std::min(m, pred_mask[i + w]);
            pred[i] = m > 60 ? 255 : 0;
        }
    memset(merged, 0, n);
    std::vector<int> order(n_cands);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scores[a] < scores[b]; });
    std::vector<int> lab, pixels;
    std::vector<Comp> comps;
    for (int ci : order) {
        const int nc = label8(cands + (size_t)ci * n, h, w, lab, comps, pixels);
        for (int l = 1; l <= nc; ++l) {
            const Comp &c = comps[l];
            if ((c.x1 - c.x0 + 1) * (c.y1 - c.y0 + 1) < 3) continue;
            try_merge(pixels.data() + c.first, c.area, pred.data(), merged);
        }
    }
    if (inpaint_dilate) {  // 5x5 rectangle, -inf outside (cv2.dilate): separable max
        std::vector<uint8_t> tmp(n);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                uint8_t m = 0;
                for (int d = -2; d <= 2; ++d)
                    if (x + d >= 0 && x + d < w) m = std::max(m, merged[y * w + x + d]);
                tmp[y * w + x] = m;
            }
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                uint8_t m = 0;
                for (int d = -2; d <= 2; ++d)
                    if (y + d >= 0 && y + d < h) m = std::max(m, tmp[(y + d) * w + x]);
                merged[y * w + x] = m;
            }
    }
    // fill holes: components of the complement below the second-largest complement area (textmask.py:113-131); the area list
    // includes "label 0" = the pixels of the mask itself, like cv2.connectedComponentsWithStats' background row
    std::vector<uint8_t> inv(n);
    int zero_area = 0;
    for (int i = 0; i < n; ++i) {
        inv[i] = merged[i] ? 0 : 255;
        zero_area += merged[i] ? 1 : 0;
    }
    const int nc = label8(inv.data(), h, w, lab, comps, pixels);
    std::vector<int> areas(1, zero_area);
    for (int l = 1; l <= nc; ++l) areas.push_back(comps[l].area);
    std::sort(areas.begin(), areas.end());
    const int thresh = areas.size() > 1 ? areas[areas.size() - 2] : areas.back();
    for (int l = 1; l <= nc; ++l)  // label 0 is the mask itself: merging it changes nothing
        if (comps[l].area < thresh) try_merge(pixels.data() + comps[l].first, comps[l].area, pred.data(), merged);
    return 0;
}

// OpenCV's getThreshVal_Otsu_8u on 256-bin histograms (cv2.threshold(..., THRESH_OTSU) of get_otsuthresh_masklist,
// ctd_utils/textmask.py:44-54): the grey level maximising the between-class variance, first maximum, same double-precision
// recurrence as hostglue._otsu_threshold — evaluated here for the (3 channels x lines) histograms the GPU refine_mask hands back,
// where the Python loop cost more than the GPU phases.
extern "C" int mit_otsu_from_hist(const int32_t *hist, int n, int32_t *thresholds) {
    if (!hist || !thresholds || n < 0) return mit_set_error("mit_otsu_from_hist: bad arguments");
    for (int h = 0; h < n; ++h) {
        const int32_t *hp = hist + (size_t)h * 256;
        int64_t size = 0;
        for (int i = 0; i < 256; ++i) size += hp[i];
        int best = 0;
        if (size > 0) {
            const double scale = 1.0 / (double)size;
            double mu = 0.0;
            for (int i = 0; i < 256; ++i) mu += (double)i * (double)hp[i];
            mu *= scale;
            double q1 = 0.0, mu1 = 0.0, best_sigma = 0.0;
            for (int i = 0; i < 256; ++i) {
                const double p_i = (double)hp[i] * scale;
                mu1 *= q1;
                q1 += p_i;
                const double q2 = 1.0 - q1;
                const double mn = q1 < q2 ? q1 : q2, mx = q1 < q2 ? q2 : q1;
                if (mn < 2.220446049250313e-16 || mx > 1.0 - 2.220446049250313e-16) continue;
                mu1 = (mu1 + (double)i * p_i) / q1;
                const double mu2 = (mu - q1 * mu1) / q2;
                const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
                if (sigma > best_sigma) {
                    best_sigma = sigma;
                    best = i;
                }
            }
        }
        thresholds[h] = best;
    }
    return 0;
}
