package fi.iki.yak.ts.compression.gorilla.gpu;

import java.nio.ByteBuffer;

/**
 * Bytes written by a ByteBufferBitOutput (G/ByteBufferBitInput.java:20-27).  They are packed big-endian
 * into the words a LongArrayInput would hold, plus the one word the word reader loads ahead.
 */
public class ByteBufferBitInput extends BitInput {
    private final long[] array;

    public ByteBufferBitInput(ByteBuffer buf) {
        ByteBuffer b = buf.duplicate();
        int n = b.remaining();
        array = new long[(n + 7) / 8 + 1];
        for (int i = 0; i < n; i++) array[i >> 3] |= (b.get() & 0xFFL) << (56 - 8 * (i & 7));
    }

    public ByteBufferBitInput(byte[] input) { this(ByteBuffer.wrap(input)); }

    @Override
    long[] words() { return array; }
}
