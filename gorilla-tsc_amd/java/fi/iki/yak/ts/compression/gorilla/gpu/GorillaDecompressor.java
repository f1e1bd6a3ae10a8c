package fi.iki.yak.ts.compression.gorilla.gpu;

/** Same surface as the reference's GorillaDecompressor: readPair() until it returns null. */
public class GorillaDecompressor {
    private final long[] image;
    private final int format;
    private long[][] pairs;
    private int next;

    public GorillaDecompressor(LongArrayInput input) { this(input, Gtsc.FMT_V2); }

    GorillaDecompressor(LongArrayInput input, int format) {
        this.image = input.array();
        this.format = format;
    }

    /** The next pair, or null at the end of the block.  The whole block is decoded by the first call. */
    public Pair readPair() {
        if (pairs == null) pairs = GpuBatch.decode(format, image);
        if (next >= pairs[0].length) return null;
        Pair p = new Pair(pairs[0][next], pairs[1][next]);
        next++;
        return p;
    }

    public static int decodeZigZag32(final int n) { return (n >>> 1) ^ -(n & 1); }
}
