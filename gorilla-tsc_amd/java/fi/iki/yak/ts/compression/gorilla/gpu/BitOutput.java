package fi.iki.yak.ts.compression.gorilla.gpu;

/**
 * Where a compressor's bit stream goes: the reference's interface, method for method (G/BitOutput.java:8-32).
 * Any implementation may be handed to a compressor.  The codec runs on the GPU, so the stream does not arrive
 * call by call as the points are added: when the block is coded, its bits are written through these methods in
 * order -- writeBits(word, 64) per whole word, the remaining bits in one more call -- and flush() follows when the
 * block is closed, as in the reference's close() (G/GorillaCompressor.java:84-89).
 *
 * Deployed beside the reference jar, this interface is the one to drop: import
 * fi.iki.yak.ts.compression.gorilla.BitOutput instead and the reference's own containers (or a caller's) are served.
 */
public interface BitOutput {
    /** Stores a set bit. */
    void writeBit();

    /** Stores a zero bit. */
    void skipBit();

    /** Writes the low {@code bits} bits of {@code value}, most significant first. */
    void writeBits(long value, int bits);

    /** Flushes whatever the container holds back. */
    void flush();
}
