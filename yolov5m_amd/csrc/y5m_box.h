// Box arithmetic shared by the IoU entry points and the fused loss kernels.
// Restates reference utils/bboxes_utils.py:33-87 (midpoint format) in fp32, same op order.
#pragma once
#include "y5m_common.h"

struct BoxFwd {
    float b1x1, b1y1, b1x2, b1y2, b2x1, b2y1, b2x2, b2y2;
    float w1, h1, w2, h2, dx, dy, iw, ih, inter, uni, iou, cw, ch, c_area, out;
};

// a = (x,y,w,h) prediction, b = (x,y,w,h) label
__device__ __forceinline__ BoxFwd box_iou_fwd(const float a[4], const float b[4], bool giou, float eps) {
    BoxFwd r;
    r.b1x1 = a[0] - a[2] / 2.0f;  r.b1y1 = a[1] - a[3] / 2.0f;      // :53-54
    r.b1x2 = a[0] + a[2] / 2.0f;  r.b1y2 = a[1] + a[3] / 2.0f;      // :55-56
    r.b2x1 = b[0] - b[2] / 2.0f;  r.b2y1 = b[1] - b[3] / 2.0f;      // :57-58
    r.b2x2 = b[0] + b[2] / 2.0f;  r.b2y2 = b[1] + b[3] / 2.0f;      // :59-60
    r.w1 = r.b1x2 - r.b1x1; r.h1 = r.b1y2 - r.b1y1;                 // :72
    r.w2 = r.b2x2 - r.b2x1; r.h2 = r.b2y2 - r.b2y1;
    r.dx = fminf(r.b1x2, r.b2x2) - fmaxf(r.b1x1, r.b2x1);           // :74
    r.dy = fminf(r.b1y2, r.b2y2) - fmaxf(r.b1y1, r.b2y1);           // :75
    r.iw = r.dx < 0.0f ? 0.0f : r.dx;                               // .clamp(0)
    r.ih = r.dy < 0.0f ? 0.0f : r.dy;
    r.inter = r.iw * r.ih;
    r.uni = r.w1 * r.h1 + r.w2 * r.h2 - r.inter + eps;              // :78
    r.iou = r.inter / r.uni;                                        // :80
    r.out = r.iou;
    if (giou) {
        r.cw = fmaxf(r.b1x2, r.b2x2) - fminf(r.b1x1, r.b2x1);       // :83
        r.ch = fmaxf(r.b1y2, r.b2y2) - fminf(r.b1y1, r.b2y1);       // :84
        r.c_area = r.cw * r.ch + eps;                               // :85
        r.out = r.iou - (r.c_area - r.uni) / r.c_area;              // :86
    }
    return r;
}

// reverse mode of the above: g = d/d(out). ga/gb receive d/d(a), d/d(b) (x,y,w,h).
// min/max ties split the gradient in half (torch.minimum/maximum autograd), clamp passes at 0.
__device__ __forceinline__ void box_iou_bwd(const BoxFwd& r, bool giou, float g, float ga[4], float gb[4]) {
    float d_iou = g, d_uni = 0.f, d_inter = 0.f, d_cw = 0.f, d_ch = 0.f;
    if (giou) {
        float d_carea = -g * r.uni / (r.c_area * r.c_area);
        d_uni += g / r.c_area;
        d_cw = d_carea * r.ch;
        d_ch = d_carea * r.cw;
    }
    d_inter += d_iou / r.uni;
    d_uni += -d_iou * r.inter / (r.uni * r.uni);
    float d_w1 = d_uni * r.h1, d_h1 = d_uni * r.w1, d_w2 = d_uni * r.h2, d_h2 = d_uni * r.w2;
    d_inter += -d_uni;
    float d_iw = d_inter * r.ih, d_ih = d_inter * r.iw;
    float d_dx = r.dx >= 0.0f ? d_iw : 0.0f;
    float d_dy = r.dy >= 0.0f ? d_ih : 0.0f;
    float g1x1 = 0.f, g1y1 = 0.f, g1x2 = 0.f, g1y2 = 0.f, g2x1 = 0.f, g2y1 = 0.f, g2x2 = 0.f, g2y2 = 0.f;
#define Y5M_SPLIT_MIN(A, B, GA, GB, D) { if ((A) < (B)) GA += (D); else if ((A) > (B)) GB += (D); else { GA += 0.5f * (D); GB += 0.5f * (D); } }
#define Y5M_SPLIT_MAX(A, B, GA, GB, D) { if ((A) > (B)) GA += (D); else if ((A) < (B)) GB += (D); else { GA += 0.5f * (D); GB += 0.5f * (D); } }
    Y5M_SPLIT_MIN(r.b1x2, r.b2x2, g1x2, g2x2, d_dx)
    Y5M_SPLIT_MAX(r.b1x1, r.b2x1, g1x1, g2x1, -d_dx)
    Y5M_SPLIT_MIN(r.b1y2, r.b2y2, g1y2, g2y2, d_dy)
    Y5M_SPLIT_MAX(r.b1y1, r.b2y1, g1y1, g2y1, -d_dy)
    if (giou) {
        Y5M_SPLIT_MAX(r.b1x2, r.b2x2, g1x2, g2x2, d_cw)
        Y5M_SPLIT_MIN(r.b1x1, r.b2x1, g1x1, g2x1, -d_cw)
        Y5M_SPLIT_MAX(r.b1y2, r.b2y2, g1y2, g2y2, d_ch)
        Y5M_SPLIT_MIN(r.b1y1, r.b2y1, g1y1, g2y1, -d_ch)
    }
#undef Y5M_SPLIT_MIN
#undef Y5M_SPLIT_MAX
    g1x2 += d_w1; g1x1 -= d_w1; g1y2 += d_h1; g1y1 -= d_h1;
    g2x2 += d_w2; g2x1 -= d_w2; g2y2 += d_h2; g2y1 -= d_h2;
    ga[0] = g1x1 + g1x2; ga[1] = g1y1 + g1y2;
    ga[2] = 0.5f * (g1x2 - g1x1); ga[3] = 0.5f * (g1y2 - g1y1);
    gb[0] = g2x1 + g2x2; gb[1] = g2y1 + g2y2;
    gb[2] = 0.5f * (g2x2 - g2x1); gb[3] = 0.5f * (g2y2 - g2y1);
}
