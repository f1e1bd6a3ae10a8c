package fi.iki.yak.ts.compression.gorilla.gpu;

/**
 * The library's own outputs also know the compressor that writes into them, so that their contents can be asked
 * for at any time: a block that is still open is then coded as far as it goes and the bits not yet delivered are
 * written (the snapshot getLongArray() returns before close(), T/EncodeGorillaTest.java:483-500).
 */
public abstract class TrackedOutput implements BitOutput {
    BlockEncoder writer;   // the compressor writing into this output

    void bringUpToDate() {
        if (writer != null) writer.bringOutputUpToDate();
    }
}
