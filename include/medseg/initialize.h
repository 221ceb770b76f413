// Engine + log lifecycle.  Same names, arguments and return conventions as the reference's include/initialize.h:12-21;
// the TensorRT types it leaked (nvinfer1::ICudaEngine*, the g_runtime/g_engine unique_ptrs, :15, :24-25) are replaced by
// the opaque C-ABI handle of include/mi_unet.h.
#pragma once
#include <fstream>
#include <string>

#include "../mi_unet.h"

namespace MedicalSeg {

// `trt_cache_path` keeps its name; it is now the path of the MIUNETW1 weight file (miunet/spec.py).  Creates
// <log_dir>/segmentation_log.txt (truncating), logs the reference's banner lines, returns false on any failure.
bool initialize_engine(const std::string &trt_cache_path, const std::string &log_dir);

mi_unet_t *get_engine();                    // the first device's handle (the reference's get_engine(), initialize.h:15)
mi_unet_group_t *get_engine_group();        // every device's handle (directory mode shards over them)
// The calling thread's own execution context, created on first use (the reference's get_thread_local_context(),
// include/process.h:26): a clone of get_engine() that shares its weights.  Throws when no engine is initialised.
mi_unet_t *get_thread_local_context();
std::ofstream &get_log_file();
std::string get_log_path();

}  // namespace MedicalSeg
