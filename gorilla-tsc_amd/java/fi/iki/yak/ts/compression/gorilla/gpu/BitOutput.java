package fi.iki.yak.ts.compression.gorilla.gpu;

/**
 * Where a compressor's bit stream goes (the role of the reference's BitOutput interface,
 * G/BitOutput.java:8-32).  The codec runs on the GPU, so nothing is written bit by bit: the stream
 * arrives as a word image plus its length in bits when the block is coded, and the concrete classes
 * present it the way the reference's containers do.
 */
public abstract class BitOutput {
    private long[] words = new long[0];
    private long nbits;
    private boolean closed;
    BlockEncoder writer;   // the compressor writing into this output

    void set(long[] image, long bits, boolean blockClosed) {
        words = image;
        nbits = bits;
        closed = blockClosed;
    }

    /** An output read while its compressor is still open gets the snapshot the reference returns there. */
    private void bringUpToDate() {
        if (writer != null) writer.bringOutputUpToDate();
    }

    long[] stream() { bringUpToDate(); return words; }

    long bits() { bringUpToDate(); return nbits; }

    boolean blockClosed() { return closed; }
}
