package fi.iki.yak.ts.compression.gorilla.gpu;

public class Pair {
    private final long timestamp, value;

    public Pair(long timestamp, long value) {
        this.timestamp = timestamp;
        this.value = value;
    }

    public long getTimestamp() { return timestamp; }

    public double getDoubleValue() { return Double.longBitsToDouble(value); }

    public long getLongValue() { return value; }
}
