package fi.iki.yak.ts.compression.gorilla.gpu;

import java.util.Arrays;

/** A long[] of the stream, word for word what the reference's LongArrayOutput holds (G/LongArrayOutput.java:44-145). */
public class LongArrayOutput extends TrackedOutput {
    private long[] done = new long[256];
    private int position;        // completed words
    private long current;        // the word being filled, bits from the top
    private int bitsLeft = Long.SIZE;

    public LongArrayOutput() { }

    public LongArrayOutput(int initialSize) { done = new long[Math.max(initialSize, 1)]; }

    @Override
    public void writeBit() { writeBits(1, 1); }

    @Override
    public void skipBit() { writeBits(0, 1); }

    @Override
    public void writeBits(long value, int bits) {
        if (bits < Long.SIZE) value &= (1L << bits) - 1;
        while (bits > 0) {
            int take = Math.min(bits, bitsLeft);
            long piece = take == Long.SIZE ? value : (value >>> (bits - take)) & ((1L << take) - 1);
            current |= take == Long.SIZE ? piece : piece << (bitsLeft - take);
            bitsLeft -= take;
            bits -= take;
            if (bitsLeft == 0) flipWord();
        }
    }

    /** Stores the current word whatever it holds (G/LongArrayOutput.java:136-139). */
    @Override
    public void flush() { flipWord(); }

    private void flipWord() {
        if (position == done.length) done = Arrays.copyOf(done, done.length * 2);
        done[position++] = current;
        current = 0;
        bitsLeft = Long.SIZE;
    }

    /** The words completed so far plus the current partial word: after close() the whole block and a zero word. */
    public long[] getLongArray() {
        bringUpToDate();
        long[] copy = Arrays.copyOf(done, position + 1);
        copy[position] = current;
        return copy;
    }
}
