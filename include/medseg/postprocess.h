// Reference: src/postprocess.cpp:47 (header-less there; textually included at src/process.cpp:9).
#pragma once
#include "image.h"

// hole fill (components of mask != 2 that touch no image edge and are smaller than 6 % of the image) -> 3x3 open of
// (mask == 2) -> keep 8-connected components of at least 6 % of the image -> output in {0, 2}.
medseg::Image8 postprocess_mask(const medseg::Image8 &src);
