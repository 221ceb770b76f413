// Per-image pipeline.  Reference: include/process.h:26-30, src/process.cpp:123-262.
#pragma once
#include <string>
#include <vector>

#include "image.h"

namespace MedicalSeg {

// RAW16 -> <base>_normalized.png + <base>_original_sizes.json -> UNet label map -> postprocess_mask ->
// <base>_mask.png -> <base>_contour_overlay.png + <base>.json.  Returns false (message on stderr and in the log) on failure.
bool process_single_image(const std::string &raw_path, int width, int height, const std::string &output_dir);

// N images in one device call (the reference's directory mode loops process_single_image, src/main.cpp:148-164): min/max,
// resample, quantise, UNet and argmax run on the GPU for the whole batch, the per-image artefacts and the CPU tail follow.
// Returns the number of images that succeeded; failures are reported like process_single_image's.
int process_image_batch(const std::vector<std::string> &raw_paths, const std::vector<int> &widths,
                        const std::vector<int> &heights, const std::string &output_dir);

// The device seam (src/process.cpp:123-175): 8-bit tile -> class-index map through mi_unet_infer_u8.
// Throws std::runtime_error("Inference failed: ...") like the reference.
medseg::Image8 execute_inference(const medseg::Image8 &gray_img);

// Batched form of the same seam for directory mode (src/main.cpp:148-164 loops files one by one): N tiles, one call.
std::vector<medseg::Image8> execute_inference_batch(const std::vector<medseg::Image8> &gray_imgs);

// 0/1/2 -> 0/128/255 (src/process.cpp:178-185)
medseg::Image8 mask_to_image(const medseg::Image8 &mask);

}  // namespace MedicalSeg
