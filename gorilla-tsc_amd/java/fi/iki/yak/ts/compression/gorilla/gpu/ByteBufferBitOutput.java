package fi.iki.yak.ts.compression.gorilla.gpu;

import java.nio.ByteBuffer;

/**
 * The same stream byte by byte, as the reference's ByteBufferBitOutput holds it (G/ByteBufferBitOutput.java:20-106):
 * byte i is stream bits 8i..8i+7.  Every write is masked to its width: the reference's unmasked short write
 * (G/ByteBufferBitOutput.java:79-83), which damages its own 1.x streams, is not reproduced.
 */
public class ByteBufferBitOutput extends TrackedOutput {
    private ByteBuffer bb;
    private byte b;
    private int bitsLeft = Byte.SIZE;

    public ByteBufferBitOutput() { this(4096); }

    public ByteBufferBitOutput(int initialSize) { bb = ByteBuffer.allocateDirect(Math.max(initialSize, 16)); }

    @Override
    public void writeBit() { writeBits(1, 1); }

    @Override
    public void skipBit() { writeBits(0, 1); }

    @Override
    public void writeBits(long value, int bits) {
        for (int i = bits - 1; i >= 0; i--) {
            b |= (byte) (((value >>> i) & 1L) << (bitsLeft - 1));
            if (--bitsLeft == 0) flipByte();
        }
    }

    /** Puts the current byte whatever it holds (G/ByteBufferBitOutput.java:93-96). */
    @Override
    public void flush() { flipByte(); }

    private void flipByte() {
        if (!bb.hasRemaining()) {
            ByteBuffer larger = ByteBuffer.allocateDirect(bb.capacity() * 2);
            bb.flip();
            larger.put(bb);
            bb = larger;
        }
        bb.put(b);
        b = 0;
        bitsLeft = Byte.SIZE;
    }

    /**
     * The live buffer, its position where the reference's would be: floor(N/8)+1 bytes after close() (flush() stores
     * the partial byte), floor(N/8) while the block is open.  The caller flips it.
     */
    public ByteBuffer getByteBuffer() {
        bringUpToDate();
        return bb;
    }
}
