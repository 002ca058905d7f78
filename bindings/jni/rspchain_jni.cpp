// Thin JNI shim over the C ABI (include/rspchain.h) for a Scala/JVM host.
// NOT BUILT IN THIS PIPELINE: the image has no JDK (no jni.h, javac, scala, sbt).  Every
// function is a one-line forward; all logic stays under the C ABI, which is what the tests
// exercise.  Build where a JDK exists:
//   g++ -O2 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I include \
//       bindings/jni/rspchain_jni.cpp -L rsp-chains_amd -lrspchain -o librspchain_jni.so
#if __has_include(<jni.h>)
#include <jni.h>

#include "rspchain.h"

extern "C" {

// params arrive as a direct ByteBuffer holding an rsp_chain_params (little-endian, filled by
// RspChainNative.scala with the same field order as the header).
JNIEXPORT jlong JNICALL Java_rspChain_RspChainNative_create(JNIEnv* env, jclass, jobject paramsBuf) {
  const auto* p = static_cast<const rsp_chain_params*>(env->GetDirectBufferAddress(paramsBuf));
  rsp_chain* h = nullptr;
  if (rsp_chain_create(p, &h) != RSP_OK) {
    env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), rsp_last_error());
    return 0;
  }
  return reinterpret_cast<jlong>(h);
}

JNIEXPORT void JNICALL Java_rspChain_RspChainNative_destroy(JNIEnv*, jclass, jlong h) {
  rsp_chain_destroy(reinterpret_cast<rsp_chain*>(h));
}

// memWriteWord(addr, value): FftMagCfarChainTester.scala:82-132
JNIEXPORT void JNICALL Java_rspChain_RspChainNative_memWriteWord(JNIEnv* env, jclass, jlong h, jint addr, jint value) {
  if (rsp_chain_write_reg(reinterpret_cast<rsp_chain*>(h), (uint32_t)addr, (uint32_t)value) != RSP_OK)
    env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), rsp_last_error());
}

// stream in n_frames frames from a direct ByteBuffer, words out into another (zero copy on the JVM side)
JNIEXPORT void JNICALL Java_rspChain_RspChainNative_process(JNIEnv* env, jclass, jlong h, jobject inBuf,
                                                            jlong nFrames, jobject outBuf) {
  const void* in = env->GetDirectBufferAddress(inBuf);
  auto* out = static_cast<uint32_t*>(env->GetDirectBufferAddress(outBuf));
  const int rc = rsp_chain_process(reinterpret_cast<rsp_chain*>(h), in, (size_t)nFrames, out);
  if (rc == RSP_ERR_INVALID) env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), rsp_last_error());
  else if (rc != RSP_OK) env->ThrowNew(env->FindClass("java/lang/RuntimeException"), rsp_last_error());
}

// Stream buffers in pinned host memory (rsp_host_alloc): rsp_chain_process DMAs them in place, at the PCIe link's
// rate; a plain ByteBuffer.allocateDirect is pageable and goes through the library's pinned staging ring instead.
JNIEXPORT jobject JNICALL Java_rspChain_RspChainNative_allocPinned(JNIEnv* env, jclass, jint device, jlong bytes) {
  void* p = nullptr;
  if (rsp_host_alloc((int)device, &p, (size_t)bytes) != RSP_OK) {
    env->ThrowNew(env->FindClass("java/lang/OutOfMemoryError"), rsp_last_error());
    return nullptr;
  }
  return env->NewDirectByteBuffer(p, bytes);
}

JNIEXPORT void JNICALL Java_rspChain_RspChainNative_freePinned(JNIEnv* env, jclass, jobject buf) {
  (void)rsp_host_free(env->GetDirectBufferAddress(buf));
}

}  // extern "C"
#endif
