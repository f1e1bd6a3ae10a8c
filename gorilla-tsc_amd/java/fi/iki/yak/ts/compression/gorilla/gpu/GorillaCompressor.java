package fi.iki.yak.ts.compression.gorilla.gpu;

import java.util.Arrays;

/**
 * Same constructor and methods as fi.iki.yak.ts.compression.gorilla.GorillaCompressor
 * (addValue(long,long), addValue(long,double), close()), coded on the GPU.
 * With format FMT_V1 it stands in for the 1.x Compressor.
 */
public class GorillaCompressor {
    public static final int FIRST_DELTA_BITS = 27;

    final long blockTimestamp;
    final LongArrayOutput out;
    final int format;
    private final GpuBatch batch;
    long[] ts = new long[128], val = new long[128];
    int size;
    private boolean closed;

    public GorillaCompressor(long timestamp, LongArrayOutput output) { this(timestamp, output, null, Gtsc.FMT_V2); }

    public GorillaCompressor(long timestamp, LongArrayOutput output, GpuBatch batch) { this(timestamp, output, batch, Gtsc.FMT_V2); }

    GorillaCompressor(long timestamp, LongArrayOutput output, GpuBatch batch, int format) {
        this.blockTimestamp = timestamp;
        this.out = output;
        this.batch = batch;
        this.format = format;
    }

    public void addValue(long timestamp, long value) {
        if (size == ts.length) {
            ts = Arrays.copyOf(ts, size * 2);
            val = Arrays.copyOf(val, size * 2);
        }
        ts[size] = timestamp;
        val[size++] = value;
    }

    public void addValue(long timestamp, double value) { addValue(timestamp, Double.doubleToRawLongBits(value)); }

    /** Ends the block.  Without a batch it is coded here; with one, on GpuBatch.flush(). */
    public void close() {
        if (closed) return;
        closed = true;
        GpuBatch b = batch != null ? batch : new GpuBatch();
        b.add(this);
        if (batch == null) b.flush();
    }

    public static int encodeZigZag32(final int n) { return (n << 1) ^ (n >> 31); }
}
