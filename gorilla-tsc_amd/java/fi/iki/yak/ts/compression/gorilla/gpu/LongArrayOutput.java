package fi.iki.yak.ts.compression.gorilla.gpu;

/** The block image, as LongArrayOutput.getLongArray() of the reference returns it (G/LongArrayOutput.java:141-145). */
public class LongArrayOutput extends BitOutput {
    public LongArrayOutput() { }

    public LongArrayOutput(int initialSize) { }

    /** After close() the whole block; before it the words completed so far plus the current partial word. */
    public long[] getLongArray() { return stream().clone(); }
}
