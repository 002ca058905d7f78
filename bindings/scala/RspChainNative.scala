// Reference-side binding a maintainer of milovanovic/rsp-chains would add (package rspChain).
// NOT COMPILED IN THIS PIPELINE (no JVM).  It lets the SAME FftMagCfarVanillaParameters object
// that drives the Chisel generators (src/main/scala/FftMagCfarChain.scala:21-29) drive the GPU
// path, and gives the tester's two verbs -- memWriteWord and "stream a frame" -- a GPU target.
package rspChain

import java.nio.{ByteBuffer, ByteOrder}
import chisel3.experimental.FixedPoint
import fft._
import magnitude._
import cfar._

object RspChainNative {
  System.loadLibrary("rspchain_jni")
  @native def create(params: ByteBuffer): Long
  @native def destroy(handle: Long): Unit
  @native def memWriteWord(handle: Long, addr: Int, value: Int): Unit
  @native def process(handle: Long, in: ByteBuffer, nFrames: Long, out: ByteBuffer): Unit
  /** direct ByteBuffer over pinned host memory (rsp_host_alloc): process() then moves it over PCIe in place */
  @native def allocPinned(device: Int, bytes: Long): ByteBuffer
  @native def freePinned(buf: ByteBuffer): Unit

  private def bp(t: FixedPoint): Int = t.binaryPoint.get
  private def w(t: FixedPoint): Int = t.getWidth

  /** Serialises the parameter case classes in the field order of rsp_chain_params (include/rspchain.h). */
  def marshal(p: FftMagCfarVanillaParameters, dtype: Int = 0, device: Int = 0): ByteBuffer = {
    val b = ByteBuffer.allocateDirect(512).order(ByteOrder.LITTLE_ENDIAN)
    val f = p.fftParams
    Seq(w(f.protoIQ.real.asInstanceOf[FixedPoint]), w(f.protoTwiddle.real.asInstanceOf[FixedPoint]), f.numPoints, if (f.useBitReverse) 1 else 0, if (f.runTime) 1 else 0, f.numAddPipes, f.numMulPipes).foreach(b.putInt)
    (0 until 16).foreach(i => b.putInt(if (i < f.expandLogic.length) f.expandLogic(i) else 0))
    (0 until 16).foreach(i => b.putInt(if (i < f.keepMSBorLSB.length && !f.keepMSBorLSB(i)) 0 else 1))
    b.putInt(f.minSRAMdepth); b.putInt(bp(f.protoIQ.real.asInstanceOf[FixedPoint])); b.putInt(2 /* Convergent */)
    val m = p.magParams
    Seq(w(m.protoIn.asInstanceOf[FixedPoint]), bp(m.protoIn.asInstanceOf[FixedPoint]), w(m.protoLog.get.asInstanceOf[FixedPoint]),
        bp(m.protoLog.get.asInstanceOf[FixedPoint]), m.log2LookUpWidth, if (m.useLast) 1 else 0, m.numAddPipes, m.numMulPipes).foreach(b.putInt)
    val c = p.cfarParams
    Seq(c.protoIn, c.protoThreshold, c.protoScaler).foreach { t => b.putInt(w(t)); b.putInt(bp(t)) }
    Seq(c.leadLaggWindowSize, c.guardWindowSize, if (c.sendCut) 1 else 0, c.fftSize, c.minSubWindowSize.getOrElse(-1),
        if (c.includeCASH) 1 else 0,
        c.CFARAlgorithm match { case CACFARType => 0; case GOSCFARType => 1; case GOSCACFARType => 2 },
        c.numMulPipes, 0 /* edgeMode zero */).foreach(b.putInt)
    Seq(p.fftAddress, p.magAddress, p.cfarAddress).foreach { a => b.putInt(a.base.toInt); b.putInt(a.mask.toInt) }
    Seq(p.beatBytes, dtype, device, 0, 0, 0).foreach(b.putInt)   // dopplerPoints, refDoppler, guardDoppler = 0: the 1-D chain
    Seq(0, 0).foreach(b.putInt)                                    // window, windowDoppler = RSP_WINDOW_NONE
    (0 until 6).foreach(_ => b.putInt(0))                          // reserved[6]
    b
  }
}

/** Drop-in for the DUT + BFMs of FftMagCfarChainVanillaTester (FftMagCfarChainTester.scala:34-151). */
class GpuFftMagCfarChain(params: FftMagCfarVanillaParameters) {
  private val h = RspChainNative.create(RspChainNative.marshal(params))
  def memWriteWord(addr: BigInt, value: BigInt): Unit = RspChainNative.memWriteWord(h, addr.toInt, value.toInt)
  /** axi4StreamIn: Seq[Int] as produced by RspChainTesterUtils.formAXI4StreamComplexData */
  def stream(axi4StreamIn: Seq[Int], fftSize: Int): Seq[Int] = {
    require(axi4StreamIn.length % fftSize == 0)
    val in = RspChainNative.allocPinned(0, 4L * axi4StreamIn.length).order(ByteOrder.LITTLE_ENDIAN)
    axi4StreamIn.foreach(in.putInt)
    // CFARParams.sendCut = true widens the output beat to 64 bits: two words per cell, {word, cut} (include/rspchain.h)
    val wordsPerCell = if (params.cfarParams.sendCut) 2 else 1
    val out = RspChainNative.allocPinned(0, 4L * wordsPerCell * axi4StreamIn.length).order(ByteOrder.LITTLE_ENDIAN)
    RspChainNative.process(h, in, axi4StreamIn.length / fftSize, out)
    val words = Seq.tabulate(wordsPerCell * axi4StreamIn.length)(i => out.getInt(4 * i))
    RspChainNative.freePinned(in); RspChainNative.freePinned(out)
    words
  }
  def close(): Unit = RspChainNative.destroy(h)
}
