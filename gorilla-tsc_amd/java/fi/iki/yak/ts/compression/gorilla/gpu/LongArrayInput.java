package fi.iki.yak.ts.compression.gorilla.gpu;

public class LongArrayInput {
    private final long[] array;

    public LongArrayInput(long[] array) { this.array = array; }

    long[] array() { return array; }
}
