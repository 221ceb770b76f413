// Reference: include/cleanup.h:7.
#pragma once
namespace MedicalSeg {
// Destroys the engine (device buffers, stream) and closes the log.
void cleanup_resources();
}  // namespace MedicalSeg
