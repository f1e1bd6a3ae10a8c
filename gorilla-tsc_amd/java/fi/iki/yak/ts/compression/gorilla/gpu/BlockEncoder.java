package fi.iki.yak.ts.compression.gorilla.gpu;

import java.util.Arrays;
import java.util.List;

/** What GorillaCompressor (2.x) and Compressor (1.x) share: points are buffered, the GPU codes the block. */
abstract class BlockEncoder {
    final long blockTimestamp;
    final BitOutput out;
    final int format, predictor, tableSize;
    private final GpuBatch batch;
    long[] ts = new long[128], val = new long[128];
    int size;
    private boolean closed;
    boolean coded;

    BlockEncoder(long timestamp, BitOutput output, GpuBatch batch, int format, Predictor p) {
        this.blockTimestamp = timestamp;
        this.out = output;
        this.batch = batch;
        this.format = format;
        this.predictor = p == null ? Gtsc.PRED_LAST : p.kind();
        this.tableSize = p == null ? 0 : p.tableSize();
        output.writer = this;
        output.set(new long[0], 0, false);
    }

    public void addValue(long timestamp, long value) {
        if (closed) throw new IllegalStateException("addValue after close");
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

    /** getLongArray() / getByteBuffer() before the block was coded: a snapshot of the open block, or the batch's flush. */
    void bringOutputUpToDate() {
        if (coded) return;
        if (closed) {
            if (batch != null) batch.flush();
            return;
        }
        GpuBatch.code(List.of(this), false);
    }
}
