package fi.iki.yak.ts.compression.gorilla.gpu;

/** The block image, as LongArrayOutput.getLongArray() of the reference returns it. */
public class LongArrayOutput {
    private long[] image = new long[0];

    public LongArrayOutput() { }

    public LongArrayOutput(int initialSize) { }

    void set(long[] words) { image = words; }

    public long[] getLongArray() { return image.clone(); }
}
