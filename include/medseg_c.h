/*
 * medseg_c.h -- C view of the C++ host facade (the headers under include/medseg/) so that tests and non-C++ hosts can drive the same
 * functions the reference exposes as C++ free functions.  Each entry names the reference function it mirrors.
 * All functions return 0 on success, non-zero on failure unless stated otherwise.
 */
#ifndef MEDSEG_C_H
#define MEDSEG_C_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* MedicalSeg::initialize_engine / process_single_image / cleanup_resources (include/initialize.h:12, process.h:29, cleanup.h:7) */
int medseg_initialize_engine(const char *weight_path, const char *log_dir);
int medseg_process_single_image(const char *raw_path, int width, int height, const char *output_dir);
/* MedicalSeg::process_image_batch: n RAW paths of sizes widths[i] x heights[i]; returns the number of successes */
int medseg_process_image_batch(const char *const *raw_paths, const int *widths, const int *heights, int n, const char *output_dir);
void medseg_cleanup_resources(void);
const char *medseg_get_log_path(void);

/* Preprocess::preprocess_raw (include/preprocess.h:20) and its in-memory core (src/preprocess.cpp:81-118) */
int medseg_preprocess_raw(const char *raw_path, const char *png_path, const char *json_path, int w, int h);
int medseg_resample_normalize(const uint16_t *src, int w, int h, uint8_t *dst, int out_w, int out_h);

/* postprocess_mask (src/postprocess.cpp:47) and mask_to_image (src/process.cpp:178) on w*h u8 buffers */
int medseg_postprocess_mask(const uint8_t *mask, int w, int h, uint8_t *out);
int medseg_mask_to_image(const uint8_t *mask, int w, int h, uint8_t *out);

/* Mask2Polygon::extract_contours (src/mask2polygon.cpp:29): xy receives x,y pairs; start[c]..start[c+1] delimits contour c.
 * Returns the number of contours, or -1 when a capacity is too small. */
int medseg_extract_contours(const uint8_t *mask, int w, int h, int32_t *xy, int cap_points, int32_t *start, int cap_contours);
/* map_contour_points (src/mask2polygon.cpp:41) */
void medseg_map_points(const int32_t *xy, int n, double scale_x, double scale_y, int32_t *out);
/* generate_json (src/mask2polygon.cpp:68): writes the document for the given contours */
int medseg_generate_json(const int32_t *xy, const int32_t *start, int ncontours, const char *json_path, const char *base_name,
                         int original_width, int original_height);
/* the picture create_overlay_image writes (src/mask2polygon.cpp:114-129): the grey tile replicated to B,G,R with every contour
 * drawn as a closed red polyline of thickness 1 (cv::drawContours(-1, (0,0,255), 1), LINE_8); bgr_out is w*h*3 bytes */
int medseg_draw_overlay(const uint8_t *gray, int w, int h, const int32_t *xy, const int32_t *start, int ncontours, uint8_t *bgr_out);
/* Mask2Polygon::process_single_mask (src/mask2polygon.cpp:134) */
void medseg_process_single_mask(const char *mask_path, const char *output_dir, const char *json_path, const char *original_png,
                                const char *base_name);
/* PNG helpers standing in for cv::imwrite / cv::imread: channels 1 or 3 (B,G,R) */
int medseg_write_png(const char *path, const uint8_t *data, int w, int h, int channels, int level0);
int medseg_read_png(const char *path, int as_color, uint8_t *data, int cap_bytes, int *w, int *h);

#ifdef __cplusplus
}
#endif
#endif
