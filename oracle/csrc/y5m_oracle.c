/*
 * y5m_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle), never linked into the product.
 *
 * Plain-C restatement of the integer / index-exact parts of the reference hot path:
 *
 *   orc_nms_tv012              third-party dependency torchvision==0.12.0 `torchvision.ops.nms`
 *                              (reference requirements.txt:11, call site utils/bboxes_utils.py:195).
 *                              torchvision is NOT vendored under /root/reference and NOT installed in
 *                              the build image, so this restates its published CPU algorithm
 *                              (torchvision/csrc/ops/cpu/nms_kernel.cpp @ v0.12.0):
 *                              areas=(x2-x1)*(y2-y1); order = argsort(scores, descending);
 *                              greedy; w=max(0,xx2-xx1); h=max(0,yy2-yy1); inter=w*h;
 *                              ovr=inter/(iarea+area_j-inter) in fp32; suppress iff ovr > iou_threshold
 *                              with the threshold held as double.
 *                              PARITY UNPINNED at this boundary: the reference holds no golden vector /
 *                              assertion for NMS (ultralytics_files/test_nms.py only prints timings).
 *                              The sort is defined here as STABLE (lower index first among equal
 *                              scores); torchvision 0.12 used an unstable sort, so inputs with tied
 *                              scores have no defined reference answer.
 *   orc_non_max_suppression    reference utils/bboxes_utils.py:175-209 (per-image body, lines 185-203).
 *   orc_build_targets_ultra    reference ultralytics_loss.py:122-311 (ComputeLoss.build_targets).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- stable descending argsort by score (merge sort on indices) ---------------------------- */
static void merge_sort_desc(const float *s, int64_t *idx, int64_t *tmp, int64_t n) {
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n;
            int64_t hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                /* take from the right run only if strictly greater: keeps lower index first on ties */
                if (s[idx[j]] > s[idx[i]]) tmp[k++] = idx[j++];
                else tmp[k++] = idx[i++];
            }
            while (i < mid) tmp[k++] = idx[i++];
            while (j < hi) tmp[k++] = idx[j++];
        }
        memcpy(idx, tmp, (size_t)n * sizeof(int64_t));
    }
}

/* torchvision 0.12 nms_kernel.cpp restatement. boxes: (n,4) x1,y1,x2,y2 fp32. returns #kept,
 * keep_out[0..k) = kept indices in descending-score order. */
int64_t orc_nms_tv012(const float *boxes, const float *scores, int64_t n, double iou_threshold,
                      int64_t *keep_out) {
    if (n <= 0) return 0;
    int64_t *order = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    int64_t *tmp = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    float *areas = (float *)malloc((size_t)n * sizeof(float));
    uint8_t *suppressed = (uint8_t *)calloc((size_t)n, 1);
    for (int64_t i = 0; i < n; ++i) {
        order[i] = i;
        const float *b = boxes + 4 * i;
        areas[i] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    merge_sort_desc(scores, order, tmp, n);
    int64_t num_to_keep = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        int64_t i = order[_i];
        if (suppressed[i]) continue;
        keep_out[num_to_keep++] = i;
        float ix1 = boxes[4 * i + 0], iy1 = boxes[4 * i + 1];
        float ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        float iarea = areas[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            int64_t j = order[_j];
            if (suppressed[j]) continue;
            float xx1 = ix1 > boxes[4 * j + 0] ? ix1 : boxes[4 * j + 0];
            float yy1 = iy1 > boxes[4 * j + 1] ? iy1 : boxes[4 * j + 1];
            float xx2 = ix2 < boxes[4 * j + 2] ? ix2 : boxes[4 * j + 2];
            float yy2 = iy2 < boxes[4 * j + 3] ? iy2 : boxes[4 * j + 3];
            float w = xx2 - xx1; if (!(w > 0.0f)) w = 0.0f;   /* std::max(0, xx2-xx1) */
            float h = yy2 - yy1; if (!(h > 0.0f)) h = 0.0f;
            float inter = w * h;
            float ovr = inter / (iarea + areas[j] - inter);
            if ((double)ovr > iou_threshold) suppressed[j] = 1;
        }
    }
    free(order); free(tmp); free(areas); free(suppressed);
    return num_to_keep;
}

/* reference utils/bboxes_utils.py:185-203 for ONE image.
 * in : boxes (N,6) rows [class, score, x, y, w, h]
 * out: out_rows (<=max_det,6) rows [class, score, x1, y1, x2, y2]; out_idx = row index into the
 *      ORIGINAL (N,6) input for each kept row. returns #kept (<= max_det). */
int64_t orc_non_max_suppression(const float *boxes, int64_t N, float threshold, double iou_threshold,
                                int64_t max_det, float *out_rows, int64_t *out_idx) {
    int64_t *src = (int64_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
    float *cand = (float *)malloc((size_t)(N > 0 ? N : 1) * 6 * sizeof(float));
    int64_t n = 0;
    for (int64_t i = 0; i < N; ++i) {
        if (boxes[6 * i + 1] > threshold) {            /* :186 strict >, fp32 compare */
            memcpy(cand + 6 * n, boxes + 6 * i, 6 * sizeof(float));
            src[n++] = i;
        }
    }
    float *xyxy = (float *)malloc((size_t)(n > 0 ? n : 1) * 4 * sizeof(float));
    float *sc = (float *)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    for (int64_t i = 0; i < n; ++i) {
        float *r = cand + 6 * i;
        r[2] = r[2] - (r[4] / 2.0f);                   /* :190 x1 = x - w/2 */
        r[3] = r[3] - (r[5] / 2.0f);                   /* :191 y1 = y - h/2 */
        r[5] = r[5] + r[3];                            /* :192 y2 = h + y1 */
        r[4] = r[4] + r[2];                            /* :193 x2 = w + x1 */
        xyxy[4 * i + 0] = r[2] + r[0];                 /* :195 boxes[...,2:] + class */
        xyxy[4 * i + 1] = r[3] + r[0];
        xyxy[4 * i + 2] = r[4] + r[0];
        xyxy[4 * i + 3] = r[5] + r[0];
        sc[i] = r[1];
    }
    int64_t *keep = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    int64_t k = orc_nms_tv012(xyxy, sc, n, iou_threshold, keep);
    if (k > max_det) k = max_det;                      /* :202-203 */
    for (int64_t i = 0; i < k; ++i) {
        memcpy(out_rows + 6 * i, cand + 6 * keep[i], 6 * sizeof(float));
        out_idx[i] = src[keep[i]];
    }
    free(src); free(cand); free(xyxy); free(sc); free(keep);
    return k;
}

/* reference utils/bboxes_utils.py:129-173 non_max_suppression_aladdin for ONE list.
 * in : boxes (N,6) rows [class, score, c0, c1, c2, c3] (fp32: torch.tensor(list) rounds the Python floats);
 *      midpoint != 0 <=> box_format == "midpoint"
 * out: out_idx = indices (into the input list) of the kept rows, in keep order. returns #kept.
 * :149 filter `score > threshold` on Python floats (double compare); :150 stable sort, descending score;
 * :151-152 truncate to max_det BEFORE suppression; :154-171 a chosen box removes the later boxes of the same
 * class unless intersection_over_union(chosen, box) < iou_threshold (A8, :33-87, fp32 tensors: the Python
 * threshold is compared in float32). */
static float orc_iou_a8(const float *a, const float *b, int midpoint) {
    float b1x1, b1y1, b1x2, b1y2, b2x1, b2y1, b2x2, b2y2;
    if (midpoint) {                                   /* :52-60 */
        b1x1 = a[0] - a[2] / 2.0f; b1y1 = a[1] - a[3] / 2.0f; b1x2 = a[0] + a[2] / 2.0f; b1y2 = a[1] + a[3] / 2.0f;
        b2x1 = b[0] - b[2] / 2.0f; b2y1 = b[1] - b[3] / 2.0f; b2x2 = b[0] + b[2] / 2.0f; b2y2 = b[1] + b[3] / 2.0f;
    } else {                                          /* :62-70 */
        b1x1 = a[0]; b1y1 = a[1]; b1x2 = a[2]; b1y2 = a[3];
        b2x1 = b[0]; b2y1 = b[1]; b2x2 = b[2]; b2y2 = b[3];
    }
    const float w1 = b1x2 - b1x1, h1 = b1y2 - b1y1, w2 = b2x2 - b2x1, h2 = b2y2 - b2y1;      /* :72 */
    float iw = (b1x2 < b2x2 ? b1x2 : b2x2) - (b1x1 > b2x1 ? b1x1 : b2x1);                      /* :74 */
    float ih = (b1y2 < b2y2 ? b1y2 : b2y2) - (b1y1 > b2y1 ? b1y1 : b2y1);                      /* :75 */
    iw = iw > 0.0f ? iw : 0.0f; ih = ih > 0.0f ? ih : 0.0f;
    const float inter = iw * ih;
    const float uni = w1 * h1 + w2 * h2 - inter + 1e-7f;                                        /* :78 */
    return inter / uni;                                                                         /* :80 */
}

int64_t orc_nms_aladdin(const float *boxes, int64_t N, double threshold, float iou_threshold, int midpoint,
                        int64_t max_det, int64_t *out_idx) {
    int64_t *idx = (int64_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
    int64_t *tmp = (int64_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
    float *sc = (float *)malloc((size_t)(N > 0 ? N : 1) * sizeof(float));
    int64_t n = 0;
    for (int64_t i = 0; i < N; ++i)
        if ((double)boxes[6 * i + 1] > threshold) idx[n++] = i;                                 /* :149 */
    for (int64_t i = 0; i < N; ++i) sc[i] = boxes[6 * i + 1];
    merge_sort_desc(sc, idx, tmp, n);                 /* :150 stable: equal scores keep the list order */
    if (n > max_det) n = max_det;                     /* :151-152 */
    char *dead = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (dead[i]) continue;
        const float *ci = boxes + 6 * idx[i];
        out_idx[k++] = idx[i];
        for (int64_t j = i + 1; j < n; ++j) {
            if (dead[j]) continue;
            const float *cj = boxes + 6 * idx[j];
            if (cj[0] != ci[0]) continue;                                                       /* :163 */
            if (!(orc_iou_a8(ci + 2, cj + 2, midpoint) < iou_threshold)) dead[j] = 1;           /* :164-168 */
        }
    }
    free(idx); free(tmp); free(sc); free(dead);
    return k;
}

/* python-style remainder a % 1.0 (torch.remainder) */
static float py_mod1(float a) {
    float r = fmodf(a, 1.0f);
    if (r != 0.0f && r < 0.0f) r += 1.0f;
    return r;
}

/* ComputeLoss.build_targets for ONE scale (reference ultralytics_loss.py:162-309).
 * targets (nt,6) [img, cls, x, y, w, h] normalised; anchors (na,2) stride-divided; grid ny,nx.
 * Outputs sized for 5*na*nt rows. Row order per SURVEY B.2: offset-major, then anchor-major, then
 * target order. returns n rows. */
int64_t orc_build_targets_ultra(const float *targets, int64_t nt, const float *anchors, int64_t na,
                                int64_t ny, int64_t nx, float anchor_t,
                                int64_t *b_out, int64_t *a_out, int64_t *gj_out, int64_t *gi_out,
                                float *tbox_out, float *anch_out, int64_t *tcls_out) {
    if (nt == 0) return 0;
    const float g = 0.5f;
    const float off[5][2] = {{0, 0}, {1, 0}, {0, 1}, {-1, 0}, {0, -1}};
    int64_t cap = na * nt;
    float *t = (float *)malloc((size_t)cap * 7 * sizeof(float));   /* filtered rows */
    int64_t nf = 0;
    float gain[7] = {1, 1, (float)nx, (float)ny, (float)nx, (float)ny, 1};
    for (int64_t a = 0; a < na; ++a) {
        for (int64_t i = 0; i < nt; ++i) {
            float row[7];
            for (int c = 0; c < 6; ++c) row[c] = targets[6 * i + c] * gain[c];
            row[6] = (float)a * gain[6];
            float rw = row[4] / anchors[2 * a + 0];
            float rh = row[5] / anchors[2 * a + 1];
            float mw = fmaxf(rw, 1.0f / rw);     /* torch.max(r, 1/r) */
            float mh = fmaxf(rh, 1.0f / rh);
            /* torch.max propagates NaN; fmaxf does not: handle explicitly */
            if (isnan(rw) || isnan(1.0f / rw)) mw = NAN;
            if (isnan(rh) || isnan(1.0f / rh)) mh = NAN;
            float m = fmaxf(mw, mh);
            if (isnan(mw) || isnan(mh)) m = NAN;
            if (m < anchor_t) { memcpy(t + 7 * nf, row, sizeof(row)); ++nf; }
        }
    }
    int64_t n = 0;
    for (int o = 0; o < 5; ++o) {
        for (int64_t r = 0; r < nf; ++r) {
            const float *row = t + 7 * r;
            float gx = row[2], gy = row[3];
            float gxi = gain[2] - gx, gyi = gain[3] - gy;
            int take;
            switch (o) {
                case 0: take = 1; break;
                case 1: take = (py_mod1(gx) < g) && (gx > 1.0f); break;    /* j */
                case 2: take = (py_mod1(gy) < g) && (gy > 1.0f); break;    /* k */
                case 3: take = (py_mod1(gxi) < g) && (gxi > 1.0f); break;  /* l */
                default: take = (py_mod1(gyi) < g) && (gyi > 1.0f); break; /* m */
            }
            if (!take) continue;
            float ox = off[o][0] * g, oy = off[o][1] * g;
            int64_t gi = (int64_t)(gx - ox);     /* .long(): trunc toward zero */
            int64_t gj = (int64_t)(gy - oy);
            if (gj < 0) gj = 0;                                 /* clamp_ mutates gij (:285) */
            if (gj > ny - 1) gj = ny - 1;
            if (gi < 0) gi = 0;
            if (gi > nx - 1) gi = nx - 1;
            b_out[n] = (int64_t)row[0];
            tcls_out[n] = (int64_t)row[1];
            a_out[n] = (int64_t)row[6];
            gj_out[n] = gj; gi_out[n] = gi;
            tbox_out[4 * n + 0] = gx - (float)gi;               /* gxy - gij (clamped) :296 */
            tbox_out[4 * n + 1] = gy - (float)gj;
            tbox_out[4 * n + 2] = row[4];
            tbox_out[4 * n + 3] = row[5];
            anch_out[2 * n + 0] = anchors[2 * a_out[n] + 0];
            anch_out[2 * n + 1] = anchors[2 * a_out[n] + 1];
            ++n;
        }
    }
    free(t);
    return n;
}
