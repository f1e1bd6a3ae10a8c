package fi.iki.yak.ts.compression.gorilla.gpu;

/** What GorillaDecompressor (2.x) and Decompressor (1.x) share: the first readPair() decodes the block on the GPU. */
abstract class BlockDecoder {
    private final long[] image;
    private final int format, predictor, tableSize;
    private long[][] pairs;
    private int next;

    BlockDecoder(BitInput input, int format, Predictor p) {
        this.image = input.words();
        this.format = format;
        if (p == null || p instanceof fi.iki.yak.ts.compression.gorilla.gpu.predictors.LastValuePredictor) {
            this.predictor = Gtsc.PRED_LAST;
            this.tableSize = 0;
        } else if (p instanceof fi.iki.yak.ts.compression.gorilla.gpu.predictors.DifferentialFCM) {
            this.predictor = Gtsc.PRED_DFCM;
            this.tableSize = ((fi.iki.yak.ts.compression.gorilla.gpu.predictors.DifferentialFCM) p).tableSize();
        } else {
            throw new IllegalArgumentException("only LastValuePredictor and DifferentialFCM run on the GPU");
        }
    }

    /** The next pair, or null at the end of the block. */
    public Pair readPair() {
        if (pairs == null) pairs = GpuBatch.decode(format, predictor, tableSize, image);
        if (next >= pairs[0].length) {
            // the points before the damage were returned; now what LongArrayInput throws (G/LongArrayInput.java:44)
            if (pairs.length > 2) throw new ArrayIndexOutOfBoundsException("compressed array ended inside a value");
            return null;
        }
        Pair p = new Pair(pairs[0][next], pairs[1][next]);
        next++;
        return p;
    }
}
