package fi.iki.yak.ts.compression.gorilla.gpu.predictors;

import fi.iki.yak.ts.compression.gorilla.gpu.Predictor;

/**
 * Differential finite context method with a table of `size` entries, rounded up to a power of two
 * (G/predictors/DifferentialFCM.java:23-46).  On the GPU every block has a table of its own; update/predict here
 * answer as the reference's do for callers that use the object directly.
 */
public class DifferentialFCM implements Predictor {
    private final int size;
    private final long[] table;
    private final int mask;
    private long lastValue = 0L;
    private int lastHash = 0;

    public DifferentialFCM(int size) {
        if (size <= 0) throw new IllegalArgumentException("Size cannot be negative and should be power of 2");
        this.size = size;
        int n = 1;
        while (n < size) n <<= 1;
        table = new long[n];
        mask = n - 1;
    }

    /** The size the constructor was given: what the library is told (it rounds up the same way). */
    public int tableSize() { return size; }

    @Override
    public void update(long value) {
        table[lastHash] = value - lastValue;
        lastHash = (int) (((lastHash << 5) ^ ((value - lastValue) >> 50)) & mask);
        lastValue = value;
    }

    @Override
    public long predict() { return table[lastHash] + lastValue; }
}
