// Host stand-ins for the HIP runtime calls the orchestration code makes  --  TEST INFRASTRUCTURE ONLY.
// oracle/emu/build.sh compiles sam_audio_amd/csrc/{engine,peav,api}.hip and this file with -Dhip<Call>=emu_hip<Call>
// for exactly the calls below, so the unchanged product sources bind to these definitions: "device" pointers are
// host pointers, streams are ignored (everything runs synchronously), events record nothing.
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>

extern "C" {
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t) {
  std::memset(dst, value, bytes);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t) {
  std::memmove(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
  std::memmove(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
// SAMAUDIO_OPT_ODE_GRAPH: no capture here - the engine falls back to its eager launches
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned) { return hipErrorNotSupported; }
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = (hipEvent_t)std::malloc(8);
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  std::free((void*)e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) {
  *ms = 0.f;
  return hipSuccess;
}
}
