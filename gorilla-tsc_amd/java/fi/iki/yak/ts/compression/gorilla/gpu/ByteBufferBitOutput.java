package fi.iki.yak.ts.compression.gorilla.gpu;

import java.nio.ByteBuffer;

/**
 * The same stream byte by byte, as the reference's ByteBufferBitOutput holds it
 * (G/ByteBufferBitOutput.java:42-103): byte i is stream bits 8i..8i+7.
 */
public class ByteBufferBitOutput extends BitOutput {
    public ByteBufferBitOutput() { }

    public ByteBufferBitOutput(int initialSize) { }

    /**
     * A direct buffer whose position is where the reference's would be: floor(N/8)+1 bytes after close()
     * (flush() stores the partial byte), floor(N/8) while the block is open.  The caller flips it.
     */
    public ByteBuffer getByteBuffer() {
        long[] w = stream();
        int n = (int) (bits() / 8) + (blockClosed() ? 1 : 0);
        ByteBuffer bb = ByteBuffer.allocateDirect(Math.max(n, w.length * 8));
        for (int i = 0; i < n; i++) bb.put((byte) (w[i >> 3] >>> (56 - 8 * (i & 7))));
        return bb;
    }
}
